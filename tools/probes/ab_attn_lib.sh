cd $GRAFT_REPO_ROOT
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_new.so
for r in 1 2; do
 for w in old new; do
  if [ $w = old ]; then cp tools/probes/lib_old.so mmduet_amd/csrc/libmmduet_hip.so; else cp /tmp/lib_new.so mmduet_amd/csrc/libmmduet_hip.so; fi
  echo "== $w"; ATTN_LIBRARY=0 python3 tools/vit_attn_bench.py 5 2>&1 | grep -E "chunk attention"
 done
done
cp /tmp/lib_new.so mmduet_amd/csrc/libmmduet_hip.so
bash tools/probes/lib_ab_bench.sh tools/probes/lib_old.so 2
