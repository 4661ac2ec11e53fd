// One instantiation of the ring GEMM for register / ISA audits (seconds to compile instead of a minute for gemm.hip):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Imumduet_amd/csrc -Rpass-analysis=kernel-resource-usage -save-temps -c tools/probes/ring_probe.hip
#include "gemm_ring.h"
#ifndef PROBE_EPI
#define PROBE_EPI 0
#endif
#ifndef PROBE_DBG
#define PROBE_DBG 0
#endif
#ifndef PROBE_DEFER
#define PROBE_DEFER false
#endif
template __global__ void gemm_ringx_kernel<PROBE_EPI, 4, false, 3, true, PROBE_DBG>(GemmP, int);
