#!/bin/bash
# headline bench under different tower burst schedules / tower batch sizes (same box, back to back): "burst blocks vit_batch [frames_per_forward]"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
while read -r b k vb fpf; do
  [ -z "$b" ] && continue
  extra=""; [ -n "$fpf" ] && extra="--frames-per-forward $fpf"
  v=$(MMDUET_VIT_BURST=$b MMDUET_VIT_BURST_BLOCKS=$k python3 bench.py --steps 4 --warmup 1 --multi-stream 0 --no-cpu-baseline --vit-batch $vb $extra 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "burst=$b blocks=$k vit_batch=$vb fpf=${fpf:-default} -> $v"
done <<< "${1:-3 128 35}"
