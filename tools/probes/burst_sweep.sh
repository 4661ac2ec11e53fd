#!/bin/bash
# headline bench under different tower burst schedules (same box, back to back)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "0 128" "3 128" "2 128" "4 128" "3 96" "3 160" "0 128" "3 128"; do
  set -- $cfg
  v=$(MMDUET_VIT_BURST=$1 MMDUET_VIT_BURST_BLOCKS=$2 python3 bench.py --steps 4 --warmup 1 --multi-stream 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "burst=$1 blocks=$2 -> $v"
done
