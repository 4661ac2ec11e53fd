#!/bin/bash
# A/B of chunk-attention variants inside one box: kernel-trace averages of attn_gqa128_kernel<2,...> for MMDUET_ATTN_QK2=0/1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
  export MMDUET_ATTN_QK2=$v
  rm -rf $R/gpurun_out/prof_ab
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ab -o trace -- python3 $R/bench.py --steps 1 --warmup 0 --no-prof --no-overlap --multi-stream 0 --no-cpu-baseline > /dev/null 2>&1
  db=$(ls $R/gpurun_out/prof_ab/*.db | head -1)
  echo "== QK2=$v"; python3 $R/tools/rocpd_stats.py $db 45 | grep -E "attn_gqa128_kernel<2|TOTAL"
  rm -rf $R/gpurun_out/prof_ab
done
