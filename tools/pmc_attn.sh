#!/bin/bash
# PMC passes over tools/vit_attn_bench.py (ViT + chunk attention inside the model): per-kernel counter averages into gpurun_out/<tag>_pmc_attn_<pass>.txt
tag=${1:-attn}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
            "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf $O/pmc_$tag
  ATTN_LIBRARY=0 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$tag -o p -- python3 $R/tools/vit_attn_bench.py 1 > $O/${tag}_pmc_attn_$i.log 2>&1
  db=$(ls $O/pmc_$tag/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 $R/tools/pmc_summary.py $db 2>/dev/null > $O/${tag}_pmc_attn_$i.txt
  rm -rf $O/pmc_$tag
done
grep -A9 "attn_gqa128_kernel<2, 4, 8>\|attn_rowmajor_kernel\|attn_d72_ring_kernel" $O/${tag}_pmc_attn_1.txt $O/${tag}_pmc_attn_2.txt | cut -c1-120
