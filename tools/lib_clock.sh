#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in "8192 8192 8192" "25515 3584 3584" "1274 37888 3584"; do
  for fill in "" constant; do
    python3 $R/tools/lib_gemm_clock.py $shape $fill 2>/dev/null | tail -1
    d=$O/pmc_lib; rm -rf $d
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d $d -o p -- python3 $R/tools/lib_gemm_clock.py $shape $fill > $d.log 2>&1
    db=$(ls $d/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python3 $R/tools/pmc_clock.py $db 2>/dev/null | python3 -c "
import sys,json; d=json.load(sys.stdin)
for k,v in d.get('kernels',{}).items():
    if v.get('avg_us',0) > 80 and v.get('mfma_busy_frac_of_cycles',0) > 0.1: print('   under pmc:', k[:50], v)"
    rm -rf $d
  done
done
