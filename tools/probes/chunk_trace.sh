R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/attn_w1_probe.py chunk > $O/r05_chunk_probe_3.txt 2>&1; tail -26 $O/r05_chunk_probe_3.txt
for shape in "1274 0" "1274 15000" "1911 1024"; do
  s=${shape// /_}
  rm -rf $O/kt_c
  rocprofv3 --kernel-trace -d $O/kt_c -o p -- python3 $R/tools/one_attn.py $shape 6 30 > /dev/null 2>&1
  db=$(ls $O/kt_c/*.db 2>/dev/null | head -1)
  echo "== $shape"; [ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db 6 | cut -c1-140
  rm -rf $O/kt_c
done
