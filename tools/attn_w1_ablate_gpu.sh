#!/bin/bash
# GPU half: wall per ablation + SQ counters of variant 5 vs 3 at the chunk shape -> gpurun_out/<tag>_*
tag=${1:-r05_w1}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/${tag}_ablate.txt
python3 $R/tools/attn_w1_ablate.py libmmduet_hip.so "shipped build" >> $O/${tag}_ablate.txt 2>&1
i=0
for name in "no exp2" "no row-sum adds" "no score MFMAs" "no P.V MFMAs" "no fragment reads" "no tile barrier / DMA wait" "no maximum chain"; do
  i=$((i+1)); [ -f $R/mmduet_amd/csrc/libmmduet_hip_w1dbg$i.so ] && python3 $R/tools/attn_w1_ablate.py libmmduet_hip_w1dbg$i.so "$name" 2>&1 | grep -v amdgpu.ids >> $O/${tag}_ablate.txt
done
python3 $R/tools/attn_w1_ablate.py libmmduet_hip.so "shipped build (again)" 2>&1 | grep -v amdgpu.ids >> $O/${tag}_ablate.txt
for v in 5 3; do
  i=0
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
              "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" \
              "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf $O/pmc_$tag
    rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$tag -o p -- python3 $R/tools/one_attn.py 1274 15000 $v 10 > $O/${tag}_pmc_v${v}_$i.log 2>&1
    db=$(ls $O/pmc_$tag/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python3 $R/tools/pmc_summary.py $db attn_gqa 2>/dev/null > $O/${tag}_pmc_v${v}_$i.txt
    rm -rf $O/pmc_$tag
  done
done
cat $O/${tag}_ablate.txt; cat $O/${tag}_pmc_v5_*.txt $O/${tag}_pmc_v3_*.txt
