// What bounds a tile GEMM's epilogue burst -- every CU storing its 128 KB output tile at the same moment?  Per-CU store issue, or the chip's write bandwidth?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/store_burst_probe.hip -o /tmp/sb && /tmp/sb
// Each block (8 waves) writes `kb` KB as the ring GEMM's epilogue does: per wave-instruction 16 rows x 64 bytes (row stride = ld bytes), 16-byte stores.
// Swept: blocks (64 .. 256: a chip-level limit shows as time ~ blocks, a per-CU limit as constant time), bytes per block, row stride, plus a contiguous-1 KB form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// MAP: 0 = 1 KB contiguous per instruction; 1 = the ring GEMM's epilogue today (lane = lq * 16 + lr: row lr, 16-byte chunk lq -> 16 rows x 64 B, a row's lanes 16 apart);
//      2 = 16 rows x 64 B with a row's four lanes ADJACENT; 3 = 8 rows x 128 B (full lines), a row's eight lanes adjacent; 4 = 8 rows x 128 B, a row's lanes 8 apart;
//      5 = 8 rows x 128 B with lane = lq * 16 + lr: row lr & 7, chunk (lr >> 3) * 4 + lq (what one DPP row_ror:8 exchange per pair of stores would give the GEMM)
template <int MAP>
__global__ __launch_bounds__(512) void burst(char* out, long long ld, int stores_per_wave, int rounds, long long round_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const u32x4 v = {(unsigned)threadIdx.x, blockIdx.x, 3u, 4u};
    for (int r = 0; r < rounds; ++r) {
        char* base = out + (long long)r * round_stride + (long long)blockIdx.x * 256 * ld;          // the block's 256 output rows
        for (int s = 0; s < stores_per_wave; ++s) {
            char* p;
            const long long r0 = (wave >> 2) * 128 + (s >> 1) * 16; const int c0 = (wave & 3) * 128;          // the wave's 16-row group and its 128-byte column range
            if (MAP == 0) p = base + ((long long)wave * stores_per_wave + s) * 1024 + lane * 16;
            else if (MAP == 1) p = base + (r0 + lr) * ld + c0 + (s & 1) * 64 + lq * 16;
            else if (MAP == 2) p = base + (r0 + (lane >> 2)) * ld + c0 + (s & 1) * 64 + (lane & 3) * 16;
            else if (MAP == 3) p = base + (r0 + (s & 1) * 8 + (lane >> 3)) * ld + c0 + (lane & 7) * 16;
            else if (MAP == 4) p = base + (r0 + (s & 1) * 8 + (lane & 7)) * ld + c0 + (lane >> 3) * 16;
            else p = base + (r0 + (s & 1) * 8 + (lr & 7)) * ld + c0 + ((lr >> 3) * 4 + lq) * 16;
            *reinterpret_cast<u32x4*>(p) = v;
        }
        __syncthreads();
    }
}
int main() {
    char* buf; const size_t bytes = (size_t)3 << 30;          // 6 rounds x 256 blocks x 256 rows x 6912 B = 2.7 GB
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](int map, int blocks, int spw, long long ld, int rounds) {
        std::vector<float> t;
        const long long rs = map == 0 ? (long long)256 * 8 * spw * 1024 : (long long)256 * 256 * ld;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(a);
            switch (map) {
                case 0: hipLaunchKernelGGL(burst<0>, dim3(blocks), dim3(512), 0, 0, buf, ld, spw, rounds, rs); break;
                case 1: hipLaunchKernelGGL(burst<1>, dim3(blocks), dim3(512), 0, 0, buf, ld, spw, rounds, rs); break;
                case 2: hipLaunchKernelGGL(burst<2>, dim3(blocks), dim3(512), 0, 0, buf, ld, spw, rounds, rs); break;
                case 3: hipLaunchKernelGGL(burst<3>, dim3(blocks), dim3(512), 0, 0, buf, ld, spw, rounds, rs); break;
                case 4: hipLaunchKernelGGL(burst<4>, dim3(blocks), dim3(512), 0, 0, buf, ld, spw, rounds, rs); break;
                default: hipLaunchKernelGGL(burst<5>, dim3(blocks), dim3(512), 0, 0, buf, ld, spw, rounds, rs); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double mb = (double)blocks * 8 * spw * 1024 * rounds / 1e6;
        static const char* names[] = {"contig1KB", "gemm16x64B", "adj16x64B", "adj8x128B", "str8x128B", "dpp8x128B"};
        printf("BURST %-10s blocks %3d  %3d KB/block x %d rounds  ld %6lld B : %8.2f us  (%.2f us per round, %5.2f TB/s)\n", names[map], blocks, 8 * spw, rounds, ld,
               t[3] * 1e3, t[3] * 1e3 / rounds, mb / (t[3] * 1e3));
    };
    for (int rounds : {1, 6})
        for (int map = 0; map < 6; ++map)
            for (int blocks : {64, 256}) run(map, blocks, 16, map == 0 ? 0 : 6912, rounds);
    run(1, 256, 8, 6912, 6); run(5, 256, 8, 6912, 6);
    return 0;
}
