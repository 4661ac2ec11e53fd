#!/bin/bash
# per-kernel register / occupancy summary of one HIP source: tools/kernel_resources.sh mmduet_amd/csrc/gemm.hip [name filter]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)}
       /LDS Size/ {printf "%-60s sgpr %3s vgpr %3s agpr %3s scratch %4s occ %s lds %s\n", name, sg, vg, ag, sc, oc, $(NF-1)}' | grep -E "$filt"
