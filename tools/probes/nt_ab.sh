#!/bin/bash
# A/B of two builds of the library in one box: default vs tools-built variant (argument: path of the variant .so, e.g. mmduet_amd/csrc/libmmduet_hip_nt.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$1
cp mmduet_amd/csrc/libmmduet_hip.so /tmp/lib_default.so
for round in 1 2; do
  for which in default variant; do
    if [ $which = default ]; then cp /tmp/lib_default.so mmduet_amd/csrc/libmmduet_hip.so; else cp $V mmduet_amd/csrc/libmmduet_hip.so; fi
    echo "== $which"
    python3 tools/bench_gemm.py prod /dev/null auto 2>&1 | grep -E "M= 25515|M=  1274" | awk '{print $2, $3, $4, $8, $9, $10, $11}' | tr '\n' ';'
    echo
  done
done
cp /tmp/lib_default.so mmduet_amd/csrc/libmmduet_hip.so
