#!/bin/bash
# One round's measurement set on the GPU box (everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/):
#   tools/profile_round.sh r02
# bench lines (default = tower overlapped; --no-overlap; configs ground600 / qvh), rocprofv3 kernel-trace summary of one stream pass,
# PMC passes (MFMA busy / wave cycles; FETCH_SIZE; WRITE_SIZE -- separate runs, kernel-trace only), derived traffic JSON, per-shape GEMM table.
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
$B --steps 5 --warmup 1 > $O/${tag}_bench.json 2> $O/${tag}_bench.err
$B --steps 3 --warmup 1 --no-overlap --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_nooverlap.json 2>> $O/${tag}_bench.err
$B --steps 3 --warmup 1 --responses 0 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_noresponses.json 2>> $O/${tag}_bench.err
$B --config ground600 --steps 2 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_ground600.json 2>> $O/${tag}_bench.err
for s in 1 2 4 8; do
  $B --config qvh --streams-per-gpu $s --steps 2 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_qvh_s$s.json 2>> $O/${tag}_bench.err
done
$B --config youcook2 --steps 2 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_youcook2_fp8.json 2>> $O/${tag}_bench.err
$B --config youcook2 --streams-per-gpu 4 --steps 1 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_youcook2_fp8_s4.json 2>> $O/${tag}_bench.err
$B --config youcook2 --weights bf16 --steps 2 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_youcook2_bf16ref.json 2>> $O/${tag}_bench.err
$B --weights fp8 --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_stream300_fp8.json 2>> $O/${tag}_bench.err
$B --phase b --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_phase_b.json 2>> $O/${tag}_bench.err
$B --tower-dtype bf16 --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_tower_bf16.json 2>> $O/${tag}_bench.err
$B --config native336 --steps 3 > $O/${tag}_bench_native336.json 2>> $O/${tag}_bench.err
$B --frames-per-forward 1 --steps 2 --warmup 1 --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_fpf1.json 2>> $O/${tag}_bench.err
$B --steps 5 --warmup 1 --no-prof --multi-stream 0 --no-cpu-baseline > $O/${tag}_bench_noprof.json 2>> $O/${tag}_bench.err
P="python3 $R/bench.py --steps 1 --warmup 0 --no-prof --no-overlap --multi-stream 0 --no-cpu-baseline"
rm -rf $O/prof_$tag
rocprofv3 --kernel-trace -d $O/prof_$tag -o trace -- $P > $O/${tag}_prof.log 2>&1
db=$(ls $O/prof_$tag/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db 45 > $O/${tag}_rocprofv3_kernel_stats.txt
[ -n "$db" ] && python3 $R/tools/rocpd_window.py $db > $O/${tag}_stream_window_kernels.txt
rm -rf $O/prof_$tag
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $pass | cut -d' ' -f1)
  rm -rf $O/pmc_$tag
  rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$tag -o p -- $P > $O/${tag}_pmc_$t.log 2>&1
  db=$(ls $O/pmc_$tag/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 $R/tools/pmc_summary.py $db 2>/dev/null > $O/${tag}_pmc_$t.txt
  [ -n "$db" ] && [ "$t" = "SQ_VALU_MFMA_BUSY_CYCLES" ] && python3 $R/tools/pmc_clock.py $db 2>/dev/null > $O/${tag}_pmc_clock.json          # effective clock + MFMA-busy share of cycles per class / kernel (same dispatches)
  rm -rf $O/pmc_$tag
done
python3 $R/tools/pmc_traffic.py $O/${tag}_pmc_FETCH_SIZE.txt $O/${tag}_pmc_WRITE_SIZE.txt > $O/${tag}_pmc_traffic.json 2>> $O/${tag}_bench.err
python3 $R/tools/bench_gemm.py prod $O/${tag}_gemm_shapes.json auto,big,rx-8w-early,rx-4w-early,ring256-splitK > $O/${tag}_gemm.log 2>&1
$R/tools/gemm_shapes_clock.sh $tag > $O/${tag}_gemm_shapes_clock.log 2>&1          # -> ${tag}_gemm_shapes_auto.json (per shape: wall, clock, MFMA-busy share)
ATTN_LIBRARY=0 python3 $R/tools/vit_attn_bench.py 5 2>&1 | grep -E "ViT attention|chunk attention" > $O/${tag}_attention_shapes.txt
tail -c 1500 $O/${tag}_bench.json
