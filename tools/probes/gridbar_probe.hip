// What would a persistent decode-layer kernel buy?  A decode step is a chain of weight-streaming phases with an all-to-all dependency between them (every block of
// phase i+1 reads the whole activation vector phase i wrote).  This probe streams the four GEMV weight sets of a 7B decoder layer (33 / 26 / 272 / 136 MB) through
//   (A) one kernel per phase, the chain captured in a hipGraph (what the decode step does today),
//   (B) ONE persistent kernel per layer-chain with a grid barrier between phases,
//   (C) as B, with the first loads of the next phase issued BEFORE the barrier (the weights do not depend on the activations),
// over 28 layers of distinct weights, and reports us per layer.  Also: the bare cost of a grid barrier on 256 resident blocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gridbar_probe.hip -o tools/probes/gridbar_probe && tools/probes/gridbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int GRID = 256, THREADS = 1024, UNROLL = 4;
constexpr long long SPIN_LIMIT = 1ll << 24;

// bounded spin: a barrier that cannot complete sets *err and lets everyone through (the probe must never hang the box)
__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target, unsigned* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { *err = 1; break; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS) void bare_barrier_kernel(unsigned* cnt, int rounds, unsigned* err) {
    for (int r = 0; r < rounds; ++r) grid_barrier(cnt, (unsigned)(r + 1) * gridDim.x, err);
}

// one phase: the block's contiguous share of `n16` 16-byte words, UNROLL loads in flight per lane; the result depends on every word and on the
// activation vector `xin` (written by the previous phase's blocks), and goes to xout[block]
__device__ __forceinline__ unsigned stream_phase(const u32x4* __restrict__ w, long long n16, const unsigned* xin, u32x4* pre, bool have_pre) {
    const long long per = n16 / GRID;                          // (sizes are multiples of GRID * THREADS * UNROLL)
    const u32x4* p = w + (long long)blockIdx.x * per + threadIdx.x;
    const long long steps = per / (THREADS * UNROLL);
    unsigned acc = xin ? __hip_atomic_load(xin + (threadIdx.x & (GRID - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    long long s = 0;
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += pre[u].x ^ pre[u].y ^ pre[u].z ^ pre[u].w;
        s = 1;
    }
    for (; s < steps; ++s) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + (s * UNROLL + u) * THREADS);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    return acc;
}
__device__ __forceinline__ void prefetch_phase(const u32x4* __restrict__ w, long long n16, u32x4* pre) {
    const long long per = n16 / GRID;
    const u32x4* p = w + (long long)blockIdx.x * per + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) pre[u] = __builtin_nontemporal_load(p + (long long)u * THREADS);
}
__device__ __forceinline__ void block_store(unsigned acc, unsigned* xout) {
    __shared__ unsigned red[THREADS / 64];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned t = 0; for (int i = 0; i < THREADS / 64; ++i) t += red[i]; __hip_atomic_store(xout + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS) void phase_kernel(const u32x4* w, long long n16, const unsigned* xin, unsigned* xout) {
    u32x4 pre[UNROLL];
    block_store(stream_phase(w, n16, xin, pre, false), xout);
}

struct Chain { const u32x4* w[4]; long long n16[4]; };
template <bool PREFETCH>
__global__ __launch_bounds__(THREADS) void chain_kernel(Chain c, int layers, long long layer_stride16, unsigned* x, unsigned* cnt, unsigned* err) {
    unsigned bar = 0;
    u32x4 pre[UNROLL];
    bool have = false;
    for (int l = 0; l < layers; ++l) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const u32x4* w = c.w[ph] + (long long)l * layer_stride16;
            unsigned acc = stream_phase(w, c.n16[ph], x + ((bar & 1) ? GRID : 0), pre, have);
            block_store(acc, x + ((bar & 1) ? 0 : GRID));
            if (PREFETCH) {
                const int nph = (ph + 1) & 3; const int nl = l + (ph == 3);
                if (nl < layers) { prefetch_phase(c.w[nph] + (long long)nl * layer_stride16, c.n16[nph], pre); have = true; } else have = false;
            }
            ++bar;
            grid_barrier(cnt, bar * GRID, err);
        }
    }
}

int main() {
    const int layers = 28;
    const long long unit = (long long)GRID * THREADS * UNROLL * 16;          // 16 MiB
    const long long bytes[4] = {2 * unit, 2 * unit, 16 * unit, 8 * unit};    // 33.5 / 33.5 / 268 / 134 MB  (qkv / o / gate_up / down, rounded to the probe's granule)
    long long layer_bytes = 0; for (int i = 0; i < 4; ++i) layer_bytes += bytes[i];
    char* W; CK(hipMalloc(&W, layer_bytes * layers));
    CK(hipMemset(W, 1, layer_bytes * layers));
    unsigned *x, *cnt, *err; CK(hipMalloc(&x, 2 * GRID * 4)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(x, 0, 2 * GRID * 4)); CK(hipMemset(err, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto ms = [&](auto fn, int reps) { fn(); CK(hipStreamSynchronize(st)); float best = 1e9f; for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t; } return best; };

    // bare barrier
    for (int threads : {256, 1024}) {
        const int rounds = 2000;
        float t = ms([&] { CK(hipMemsetAsync(cnt, 0, 4, st)); hipLaunchKernelGGL(bare_barrier_kernel, dim3(GRID), dim3(threads), 0, st, cnt, rounds, err); }, 3);
        printf("bare grid barrier, %d blocks x %d threads: %.2f us per barrier\n", GRID, threads, t * 1e3 / rounds);
    }
    Chain c; long long off = 0;
    for (int i = 0; i < 4; ++i) { c.w[i] = (const u32x4*)(W + off); c.n16[i] = bytes[i] / 16; off += bytes[i]; }
    const long long stride16 = layer_bytes / 16;

    // (A) graph of one kernel per phase
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0, b = 0; l < layers; ++l) for (int ph = 0; ph < 4; ++ph, ++b)
        hipLaunchKernelGGL(phase_kernel, dim3(GRID), dim3(THREADS), 0, st, c.w[ph] + l * stride16, c.n16[ph], x + ((b & 1) ? GRID : 0), x + ((b & 1) ? 0 : GRID));
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float tA = ms([&] { CK(hipGraphLaunch(ge, st)); }, 5);
    printf("(A) graph, one kernel per phase : %.1f us per layer (%.0f MB -> %.2f TB/s)\n", tA * 1e3 / layers, layer_bytes / 1e6, layer_bytes * layers / (tA * 1e-3) / 1e12);
    float tB = ms([&] { CK(hipMemsetAsync(cnt, 0, 4, st)); hipLaunchKernelGGL(chain_kernel<false>, dim3(GRID), dim3(THREADS), 0, st, c, layers, stride16, x, cnt, err); }, 5);
    printf("(B) persistent, grid barriers   : %.1f us per layer (%.2f TB/s)\n", tB * 1e3 / layers, layer_bytes * layers / (tB * 1e-3) / 1e12);
    float tC = ms([&] { CK(hipMemsetAsync(cnt, 0, 4, st)); hipLaunchKernelGGL(chain_kernel<true>, dim3(GRID), dim3(THREADS), 0, st, c, layers, stride16, x, cnt, err); }, 5);
    printf("(C) persistent + prefetch       : %.1f us per layer (%.2f TB/s)\n", tC * 1e3 / layers, layer_bytes * layers / (tC * 1e-3) / 1e12);
    // one long phase: the streaming rate itself
    float tS = ms([&] { hipLaunchKernelGGL(phase_kernel, dim3(GRID), dim3(THREADS), 0, st, (const u32x4*)W, layer_bytes * layers / 16, (const unsigned*)nullptr, x); }, 3);
    printf("one kernel over all %.1f GB      : %.2f TB/s\n", layer_bytes * layers / 1e9, layer_bytes * layers / (tS * 1e-3) / 1e12);
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("barrier timeouts: %u\n", herr);
    return herr ? 2 : 0;
}
