#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "1 3" "0 3" "0 4" "1 3" "0 3"; do
  set -- $cfg
  v=$(MMDUET_VIT_LOOKAHEAD=$1 MMDUET_VIT_BURST=$2 python3 bench.py --steps 4 --warmup 1 --multi-stream 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "lookahead=$1 burst=$2 -> $v"
done
