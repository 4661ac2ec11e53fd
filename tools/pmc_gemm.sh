#!/bin/bash
# PMC passes on single GEMM shapes (run on the GPU box): tools/pmc_gemm.sh out_prefix "M N K epi variant" ...
# pass 1: MFMA / wave-cycle counters, pass 2: LDS counters, pass 3: FETCH_SIZE, pass 4: WRITE_SIZE  (separate runs, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shift
i=0
for shape in "$@"; do
  i=$((i+1))
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    d=$R/gpurun_out/pmc_${out}_${i}_${tag}
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass -d $d -o p -- python3 $R/tools/one_gemm.py $shape 4 > $d.log 2>&1
    db=$(ls $d/*.db 2>/dev/null | head -1)
    echo "== shape $shape  pass $tag" >> $R/gpurun_out/pmc_${out}.txt
    if [ -n "$db" ]; then python3 $R/tools/pmc_summary.py $db gemm_ring 2>/dev/null >> $R/gpurun_out/pmc_${out}.txt; else tail -5 $d.log >> $R/gpurun_out/pmc_${out}.txt; fi
    rm -rf $d
  done
done
