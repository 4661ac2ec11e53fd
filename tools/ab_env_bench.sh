#!/bin/bash
# interleaved A/B of the bench line under an environment switch: tools/ab_env_bench.sh TAG VAR=VALUE [rounds] [extra bench args]  -> gpurun_out/<TAG>_ab.txt
tag=$1; sw=$2; rounds=${3:-3}; shift; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
: > $O/${tag}_ab.txt
for r in $(seq 1 $rounds); do
  for arm in base switch; do
    if [ $arm = switch ]; then export $sw; else unset ${sw%%=*}; fi
    line=$(python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --multi-stream 0 --no-parity-check "$@" 2>/dev/null | tail -1)
    echo "$arm $(echo $line | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])')" >> $O/${tag}_ab.txt
  done
done
unset ${sw%%=*}
cat $O/${tag}_ab.txt
