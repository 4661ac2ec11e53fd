#!/bin/bash
# build-container half: one debug library per timing-only ablation of attn_gqa128_chunk_kernel (CHUNK_DBG=n, WRONG results) -> mmduet_amd/csrc/libmmduet_hip_chdbg<n>.so (git-ignored)
cd $(dirname $0)/../../mmduet_amd/csrc
for n in ${@:-0 1 2 3 4 5 6 7 8 9}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -fno-honor-nans -DCHUNK_DBG=$n -c attn.hip -o /tmp/attn_chdbg$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmmduet_hip_chdbg$n.so gemm.o /tmp/attn_chdbg$n.o ops.o model.o comm.o -ldl
done
