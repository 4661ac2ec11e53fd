cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ks in 2 3 4; do
  export MMDUET_GEMV_KSPLIT_SHORT=$ks
  rm -rf $R/gpurun_out/prof_ks
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_ks -o trace -- python3 $R/bench.py --steps 1 --warmup 0 --no-prof --no-overlap --multi-stream 0 --no-cpu-baseline > /dev/null 2>&1
  db=$(ls $R/gpurun_out/prof_ks/*.db | head -1)
  echo "== ks=$ks"; python3 $R/tools/rocpd_stats.py $db 45 | grep -E "gemv16_kernel<1, 8, false, true, 4>|slab_rope|TOTAL"
  rm -rf $R/gpurun_out/prof_ks
done
