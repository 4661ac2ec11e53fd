#!/bin/bash
# PMC passes on ONE GEMM shape for several kernel variants (A/B by counters, not wall time): tools/pmc_gemm_ab.sh out_tag "M N K epi" variant variant ...
# pass 1: MFMA / wave-cycle counters, pass 2: LDS counters, pass 3: FETCH_SIZE, pass 4: WRITE_SIZE  (separate runs, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$1; shape=$2; shift; shift
for v in "$@"; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    d=$R/gpurun_out/pmcab_${out}_${v}_${tag}
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass -d $d -o p -- python3 $R/tools/one_gemm.py $shape $v 4 > $d.log 2>&1
    db=$(ls $d/*.db 2>/dev/null | head -1)
    echo "== shape $shape  variant $v  pass $tag" >> $R/gpurun_out/pmc_${out}.txt
    if [ -n "$db" ]; then python3 $R/tools/pmc_summary.py $db gemm_ring 2>/dev/null >> $R/gpurun_out/pmc_${out}.txt; else tail -5 $d.log >> $R/gpurun_out/pmc_${out}.txt; fi
    rm -rf $d $d.log
  done
done
