#!/bin/bash
# GPU half of the chunk attention ablations (timing only, WRONG results): one library per CHUNK_DBG value (tools/probes/chunk_ablate.sh builds them), chunk attention per layer inside the model
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_keep.so
names=("shipped" "no exp2" "no row-sum adds" "no score MFMAs" "no P.V MFMAs" "no V fragment reads" "no K fragment reads" "no maximum chain" "no tile barrier / DMA wait" "no K / V staging")
: > $O/r05_chunk_ablate.txt
for n in 0 1 2 3 4 5 6 7 8 9 0; do
  cp mmduet_amd/csrc/libmmduet_hip_chdbg$n.so mmduet_amd/csrc/libmmduet_hip.so
  echo "CHUNK_DBG=$n (${names[$n]}):" >> $O/r05_chunk_ablate.txt
  ATTN_LIBRARY=0 python3 tools/vit_attn_bench.py 3 2>&1 | grep 'chunk attention' | grep -v "over 0 keys" >> $O/r05_chunk_ablate.txt
done
cp /tmp/lib_keep.so mmduet_amd/csrc/libmmduet_hip.so
cat $O/r05_chunk_ablate.txt
