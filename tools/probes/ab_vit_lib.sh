#!/bin/bash
# A/B of two builds of the library on the ViT attention inside the model, interleaved in one box: tools/probes/ab_vit_lib.sh <variant.so> [rounds]
cd $GRAFT_REPO_ROOT
V=$1; N=${2:-2}
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_new.so
for r in $(seq 1 $N); do
 for w in default variant; do
  if [ $w = variant ]; then cp $V mmduet_amd/csrc/libmmduet_hip.so; else cp /tmp/lib_new.so mmduet_amd/csrc/libmmduet_hip.so; fi
  echo "== $w: $(ATTN_LIBRARY=0 VIT_ONLY=1 python3 tools/vit_attn_bench.py 5 2>&1 | grep 'ViT attention')"
 done
done
cp /tmp/lib_new.so mmduet_amd/csrc/libmmduet_hip.so
