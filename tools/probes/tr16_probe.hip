// probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS elements does lane l receive?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    int lane = threadIdx.x;
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lds + lane * 4));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
    return 0;
}
