#!/bin/bash
# Per production GEMM shape (dispatcher's choice): wall, TF/s, effective clock and MFMA-busy share of cycles -> gpurun_out/<tag>_gemm_shapes_auto.json
# One rocprofv3 --pmc pass per shape over tools/one_gemm.py (random operands); counters and duration come from the same dispatches (tools/pmc_clock.py).
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/${tag}_gemm_shapes_clock.jsonl
while read name M N K epi; do
  d=$O/pmc_shape_$name; rm -rf $d
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d $d -o p -- python3 $R/tools/one_gemm.py $M $N $K $epi 0 8 > $d.log 2>&1
  db=$(ls $d/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    python3 $R/tools/pmc_clock.py $db gemm_ > $d.json 2>/dev/null
    python3 - "$name" $M $N $K $epi $d.json $d.log >> $O/${tag}_gemm_shapes_clock.jsonl <<'PY'
import json, sys, re
name, M, N, K, epi, path, log = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], sys.argv[7]
d = json.load(open(path))
ks = d.get('kernels', {})
main = max(ks.items(), key=lambda kv: kv[1]['avg_us'] * kv[1]['dispatches']) if ks else (None, {})
tot_us = sum(v['avg_us'] * v['dispatches'] for v in ks.values()) / max(1, main[1].get('dispatches', 1))          # helper kernels (split-K reduce) add time, not launches
tf = 2.0 * M * N * K / (tot_us * 1e-6) / 1e12 if tot_us else None
print(json.dumps(dict(name=name, M=M, N=N, K=K, epi=epi, kernel=main[0], us_under_pmc=round(tot_us, 1), tflops_under_pmc=round(tf, 1) if tf else None, frac_of_2500=round(tf / 2500, 3) if tf else None,
                      effective_clock_ghz=main[1].get('effective_clock_ghz'), mfma_busy_frac_of_cycles=main[1].get('mfma_busy_frac_of_cycles'), wait_any_frac_of_wave_cycles=main[1].get('wait_any_frac_of_wave_cycles'))))
PY
  else
    tail -3 $d.log
  fi
  rm -rf $d $d.json
done <<'SHAPES'
vit_qkv 25515 3456 1152 none
vit_o 25515 1152 1152 resid
vit_fc1 25515 4352 1152 gelu_tanh
vit_fc2 25515 1152 4352 resid
proj0 25515 3584 1152 gelu_erf
proj2 25515 3584 3584 none
llm_qkv 1274 4608 3584 none
llm_o 1274 3584 3584 resid
llm_gate_up 1274 37888 3584 swiglu
llm_down 1274 3584 18944 resid
llm_gate_up_4streams 2548 37888 3584 swiglu
llm_down_4streams 2548 3584 18944 resid
SHAPES
python3 - $O/${tag}_gemm_shapes_clock.jsonl > $O/${tag}_gemm_shapes_auto.json <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
json.dump(dict(note='tools/gemm_shapes_clock.sh: per production shape ONE rocprofv3 --kernel-trace --pmc pass over tools/one_gemm.py (dispatcher\'s choice, random bf16 operands, 8 timed + 3 warm-up '
                    'launches); wall, clock and MFMA-busy share come from the same dispatches (tools/pmc_clock.py). frac_of_2500 = TF/s / 2.5 PF (the peak at 2.4 GHz): '
                    'frac ~= mfma_busy_frac_of_cycles x (effective_clock_ghz / 2.4) x (MFMA-issue efficiency of the busy cycles)', rows=rows), sys.stdout, indent=1)
PY
cat $O/${tag}_gemm_shapes_auto.json | head -60
