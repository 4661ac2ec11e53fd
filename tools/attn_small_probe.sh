#!/bin/bash
# Anatomy of the small-S (per-frame / decode) split-KV attention: wall per shape, per-kernel durations, SQ counters, segment stamps (debug library).
# usage (GPU box): bash tools/attn_small_probe.sh <tag>        -> gpurun_out/<tag>_*
tag=${1:-r05_small}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_attn.py small > $O/${tag}_shapes.txt 2>&1
[ -f $R/mmduet_amd/csrc/libmmduet_hip_timing.so ] && python3 $R/tools/attn_small_timing.py > $O/${tag}_timing.log 2>&1
for shape in "49 15000" "1 15000"; do
  s=${shape// /_}
  rm -rf $O/kt_$tag
  rocprofv3 --kernel-trace -d $O/kt_$tag -o p -- python3 $R/tools/one_attn.py $shape 3 50 > $O/${tag}_kt_$s.log 2>&1
  db=$(ls $O/kt_$tag/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db 12 > $O/${tag}_kernels_$s.txt 2>&1
  rm -rf $O/kt_$tag
  i=0
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
              "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" \
              "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf $O/pmc_$tag
    rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$tag -o p -- python3 $R/tools/one_attn.py $shape 3 10 > $O/${tag}_pmc_${s}_$i.log 2>&1
    db=$(ls $O/pmc_$tag/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python3 $R/tools/pmc_summary.py $db attn 2>/dev/null > $O/${tag}_pmc_${s}_$i.txt
    rm -rf $O/pmc_$tag
  done
done
tail -n +1 $O/${tag}_shapes.txt $O/${tag}_timing.log $O/${tag}_kernels_*.txt | cut -c1-220
echo "--- ring form (attn_gqa128_kernel<2,4,8>) forced for rows >= 256" >> $O/${tag}_shapes.txt
python3 $R/tools/bench_attn.py small 2>&1 | grep "S=  49\|S=  64" >> $O/${tag}_shapes.txt
tail -12 $O/${tag}_shapes.txt
