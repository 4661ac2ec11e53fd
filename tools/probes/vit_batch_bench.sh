#!/bin/bash
# headline stream against the tower batch size (tile quantisation of the tower GEMMs: 256 x 256 tiles over 256 CUs), interleaved, one box: tools/probes/vit_batch_bench.sh "35 38 57 75" [rounds]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${2:-2}); do
 for b in $1; do
  python3 bench.py --vit-batch $b --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline --no-parity-check --host-frames-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('vit_batch $b:', d['value'], d['ms_per_step'], r.get('frac'), r.get('avg_launch_us'))"
 done
done
