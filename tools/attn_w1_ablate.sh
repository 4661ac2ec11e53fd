#!/bin/bash
# build-container half: one debug library per timing-only ablation of attn_gqa128_w1_kernel (W1_DBG=n, WRONG results) -> mmduet_amd/csrc/libmmduet_hip_w1dbg<n>.so (git-ignored)
cd $(dirname $0)/../mmduet_amd/csrc
for n in ${@:-1 2 3 4 5 6 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -fno-honor-nans -DW1_DBG=$n -c attn.hip -o /tmp/attn_w1dbg$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmmduet_hip_w1dbg$n.so gemm.o /tmp/attn_w1dbg$n.o ops.o model.o comm.o -ldl
done
ls -la libmmduet_hip_w1dbg*.so
