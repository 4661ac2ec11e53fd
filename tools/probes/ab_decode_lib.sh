#!/bin/bash
# A/B of two builds of the library on the decode path: 64 greedy tokens at 15 k context (tools/decode_probe.py) + the split-KV attention micro-benchmark, interleaved in one box
cd $GRAFT_REPO_ROOT
V=$1; N=${2:-2}
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_new.so
for r in $(seq 1 $N); do
 for w in default variant; do
  if [ $w = variant ]; then cp $V mmduet_amd/csrc/libmmduet_hip.so; else cp /tmp/lib_new.so mmduet_amd/csrc/libmmduet_hip.so; fi
  echo "== $w"; python3 tools/decode_probe.py 64 15000 bf16 2>&1 | grep "iter" | tail -2
  python3 tools/bench_attn.py small 2>&1 | grep "S=   1" | grep "n= 15000\|n= 30000\|n=  4096"
 done
done
cp /tmp/lib_new.so mmduet_amd/csrc/libmmduet_hip.so
