#!/bin/bash
# GPU half of the ViT attention ablations (timing only, WRONG results): one library per D72_DBG value, ViT attention per layer inside the model
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_keep.so
names=("shipped" "no exp2" "no row-sum adds" "no score MFMAs" "no P.V MFMAs" "no V fragment reads" "no K fragment reads" "no maximum chain" "no tile barrier / DMA wait")
: > $O/r05_d72_ablate.txt
for n in 0 1 2 3 4 5 6 7 8 0; do
  cp mmduet_amd/csrc/libmmduet_hip_d72dbg$n.so mmduet_amd/csrc/libmmduet_hip.so
  echo "D72_DBG=$n (${names[$n]}): $(ATTN_LIBRARY=0 VIT_ONLY=1 python3 tools/vit_attn_bench.py 5 2>&1 | grep 'ViT attention')" >> $O/r05_d72_ablate.txt
done
cp /tmp/lib_keep.so mmduet_amd/csrc/libmmduet_hip.so
cat $O/r05_d72_ablate.txt
