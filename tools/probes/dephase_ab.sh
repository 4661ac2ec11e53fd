#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 0 1 0 1 2; do
  echo "== MMDUET_RING_DEPHASE=$v"
  MMDUET_RING_DEPHASE=$v python3 tools/bench_gemm.py prod /dev/null auto 2>&1 | grep -E "M= 25515|M=  1274 gate_up|M=  1323 gate_up" | awk "{print \$3, \$9}" | tr '\n' ';'
  echo
done
