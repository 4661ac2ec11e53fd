#!/bin/bash
# bench line (bf16 + fp8) and the rocprofv3 kernel-trace summary of one stream pass -> gpurun_out/<tag>_*
tag=${1:-quick}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
$B --steps 5 --warmup 1 --no-cpu-baseline 2> $O/${tag}_bench.err | tail -1 > $O/${tag}_bench.json
$B --steps 5 --warmup 1 --no-cpu-baseline --weights fp8 2>> $O/${tag}_bench.err | tail -1 > $O/${tag}_bench_fp8.json
P="python3 $R/bench.py --steps 1 --warmup 0 --no-prof --no-overlap --multi-stream 0 --no-cpu-baseline"
rm -rf $O/prof_$tag
rocprofv3 --kernel-trace -d $O/prof_$tag -o trace -- $P > $O/${tag}_prof.log 2>&1
db=$(ls $O/prof_$tag/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db 45 > $O/${tag}_kernel_stats.txt
rm -rf $O/prof_$tag
head -c 700 $O/${tag}_bench.json
