#!/bin/bash
# build-container half: one debug library per timing-only ablation of attn_d72_ring_kernel (D72_DBG=n, WRONG results) -> mmduet_amd/csrc/libmmduet_hip_d72dbg<n>.so (git-ignored);
# GPU half: tools/probes/d72_ablate_gpu.sh
cd $(dirname $0)/../../mmduet_amd/csrc
for n in ${@:-0 1 2 3 4 5 6 7 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -fno-honor-nans -DD72_DBG=$n -c attn.hip -o /tmp/attn_d72dbg$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmmduet_hip_d72dbg$n.so gemm.o /tmp/attn_d72dbg$n.o ops.o model.o comm.o -ldl
done
ls -la libmmduet_hip_d72dbg*.so
