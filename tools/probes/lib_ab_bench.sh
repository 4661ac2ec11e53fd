#!/bin/bash
# A/B of two builds of the library inside the model, one box, interleaved: tools/probes/lib_ab_bench.sh <variant.so> [rounds] [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$1; N=${2:-3}; shift; shift
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_default.so
for round in $(seq 1 $N); do
  for which in default variant; do
    if [ $which = default ]; then cp /tmp/lib_default.so mmduet_amd/csrc/libmmduet_hip.so; else cp $V mmduet_amd/csrc/libmmduet_hip.so; fi
    python3 bench.py --steps 3 --warmup 1 --multi-stream 0 --no-cpu-baseline --no-parity-check "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$which', d['value'], d['ms_per_step'], r.get('frac'), r.get('avg_launch_us'))"
  done
done
cp /tmp/lib_default.so mmduet_amd/csrc/libmmduet_hip.so
