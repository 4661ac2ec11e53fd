// One instantiation of gemm_ringw_kernel for register / ISA audits:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Immduet_amd/csrc -S tools/probes/ringw_probe.hip
#include "gemm_ringw.h"
#ifndef PROBE_EPI
#define PROBE_EPI 0
#endif
#ifndef PROBE_NS
#define PROBE_NS 3
#endif
#ifndef PROBE_NB
#define PROBE_NB 3
#endif
template __global__ void gemm_ringw_kernel<PROBE_EPI, PROBE_NS, PROBE_NB, false>(GemmP, int);
