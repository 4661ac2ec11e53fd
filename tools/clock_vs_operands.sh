#!/bin/bash
# Does the clock of the ring GEMM follow the operand bits?  The same shape with random and with constant operands under ONE PMC pass each (tools/pmc_clock.py).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in "1274 37888 3584 swiglu" "25515 3584 3584 none" "1274 3584 18944 resid"; do
  for fill in random constant; do
    d=$O/pmc_clkop; rm -rf $d
    if [ $fill = constant ]; then export ONE_GEMM_CONSTANT=1; else unset ONE_GEMM_CONSTANT; fi
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d $d -o p -- python3 $R/tools/one_gemm.py $shape 0 8 > $d.log 2>&1
    db=$(ls $d/*.db 2>/dev/null | head -1)
    echo "== $shape operands=$fill: $(tail -1 $d.log)"
    [ -n "$db" ] && python3 $R/tools/pmc_clock.py $db gemm_ 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print({k:v for k,v in d.get('gemm_tile',{}).items()})"
    rm -rf $d
  done
done
unset ONE_GEMM_CONSTANT
