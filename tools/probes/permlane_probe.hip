// what do v_permlane16_swap / v_permlane32_swap do on gfx950?  hipcc --offload-arch=gfx950 permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = {"swap16 r0", "swap16 r1", "swap32 r0", "swap32 r1"};
    for (int s = 0; s < 4; ++s) { printf("%s:", names[s]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[s * 64 + i]); printf("\n"); }
    return 0;
}
