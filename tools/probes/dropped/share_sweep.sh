#!/bin/bash
# no-response stream (prefill only): does a capped tower grid beside the LLM prefill beat time-slicing the whole chip?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for blocks in 0 128 160 192 224 0; do
  if [ $blocks = 0 ]; then unset MMDUET_TOWER_RING MMDUET_TOWER_RING_BLOCKS; else export MMDUET_TOWER_RING=16 MMDUET_TOWER_RING_BLOCKS=$blocks; fi
  v=$(python3 bench.py --responses 0 --steps 4 --warmup 1 --multi-stream 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "tower blocks=$blocks -> $v"
done
