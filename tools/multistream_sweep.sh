#!/bin/bash
# Response-mode multi-stream sweep ("also 16 / 32 / 64 streams"): S streams per GPU in shared forwards, every stream 300 frames
# with 4 x 32-token responses at its own seeded frames.  Per configuration: frames/s, merged forwards per pass, replayed frames, share of the wall inside the forwards, and the
# GPU idle share from a rocprofv3 kernel trace of the same pass (tools/gap_trace_multi.sh).  -> gpurun_out/<tag>_multistream_sweep.json
tag=${1:-r06}; shift
CFGS=${@:-"1x26 2x26 4x13 8x6 16x3"}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
: > $O/${tag}_ms_sweep.jsonl
for cfg in $CFGS; do
  # (a) un-profiled pass: the figures; (b) profiled pass: the idle share
  python3 $R/tools/multistream_anatomy.py $cfg 1 > $O/ms_sweep_$cfg.log 2>&1
  cp $O/multistream_anatomy.json $O/ms_sweep_$cfg.json
  $R/tools/gap_trace_multi.sh $cfg 1 > /dev/null 2>&1
  python3 - $cfg $O/ms_sweep_$cfg.json $O/gap_multi_report.txt >> $O/${tag}_ms_sweep.jsonl <<'PY'
import json, re, sys
cfg, a, g = sys.argv[1], json.load(open(sys.argv[2])), open(sys.argv[3]).read()
r = a[cfg]
S, k = (int(x) for x in cfg.split('x'))
m = re.search(r'GPU busy \(union over the streams\) ([\d.]+) ms, idle ([\d.]+) ms = ([\d.]+) %', g)
print(json.dumps(dict(streams_per_gpu=S, frames_per_forward_per_stream=k, frames_per_s=r['frames_per_s'], wall_ms=r['wall_ms'], merged_forwards=r['rounds'],
                      replayed_frames=r.get('replayed'), time_in_forwards_frac=r['in_forwards'],
                      gpu_idle_frac_profiled_pass=round(float(m.group(3)) / 100, 4) if m else None, round_classes=r['classes'])))
PY
done
python3 - $O/${tag}_ms_sweep.jsonl > $O/${tag}_multistream_sweep.json <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
json.dump(dict(note='tools/multistream_sweep.sh: bench.py MultiRunner workload (S x 300-frame streams, the rank-0 frames, 4 x 32-token responses per stream at frames drawn from '
                    'random.Random(stream), bf16, fp16-autocast tower, frames resident in HBM), one warm-up + one timed pass per configuration; native decode rounds (mmd_round_multi). '
                    'round_classes: per kind of merged forward (f = a watching stream\'s frame chunk, g = a talking stream\'s row) count / time inside the forward / host time in front of it. '
                    'gpu_idle_frac from a second, rocprofv3-traced pass of the same configuration (union of both HIP streams).', rows=rows), sys.stdout, indent=1)
PY
python3 - $O/${tag}_multistream_sweep.json <<'PY'
import json, sys
for r in json.load(open(sys.argv[1]))['rows']:
    print({k: v for k, v in r.items() if k != 'round_classes'})
PY
