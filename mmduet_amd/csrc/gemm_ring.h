// gemm_ring.h -- the ring GEMM (gemm_ringx_kernel) with the parameter block and the epilogue helpers it shares with the other tile kernels.
// Included by gemm.hip (the dispatcher and every other GEMM kernel) and by tools/probes/ring_probe.hip (one instantiation, for register / ISA audits).
#pragma once
#include "common.h"
#include <type_traits>

#define MMD_BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

struct GemmP {
    const void* X; const void* W; const void* bias; const void* R; void* Y; float* ws;
    const float* wscale;   // per-output-channel weight scale (fp8-quantised matrices) or null
    long long ldx, ldw, ldr, ldy;
    int M, N, K, epi, out_f32, kper, vec;
    int slabs;         // skinny kernel: always leave fp32 slabs in ws (the consumer kernel reduces them)
    void* dump;        // ring kernel: 8 KB nobody reads -- masked output lanes store here (a per-device buffer of the launcher, never the split-K workspace)
    int flags;         // (unused)
    int x_pm, y_pm;    // ring kernel: X / Y in the PIECE-MAJOR activation layout (see gemm_ringx_kernel): piece (m >> 4, k >> 5) = 16 rows x 32 elements = 1 KB contiguous
};

// guard-free epilogue of the big-tile kernel: N % BN == 0, ldy/ldr % 4 == 0 (dispatch conditions), one 8-byte access
// per operand and tile; EPI is a compile-time constant so no per-element branches survive.
template <int EPI>
__device__ __forceinline__ s16x4_t big_value(const GemmP& p, int m, int n, const f32x4_t& a) {      // m < p.M
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.wscale) {
        const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(p.wscale + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= sc[r];
    }
    if (p.bias) {
        s16x4_t b = *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf2f((bf16_t)b[r]);
    }
    if constexpr (EPI == EPI_GELU_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(bf2f(f2bf(v[r])));
    } else if constexpr (EPI == EPI_GELU_ERF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(bf2f(f2bf(v[r])));
    } else if constexpr (EPI == EPI_RESID) {
        s16x4_t rr = *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.R + (long long)m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bf2f(f2bf(v[r])) + bf2f((bf16_t)rr[r]);
    }
    return s16x4_t{(short)f2bf(v[0]), (short)f2bf(v[1]), (short)f2bf(v[2]), (short)f2bf(v[3])};
}
// the same value from operands the caller already holds: per-column weight scale / bias quads (one load per TILE, not per row) and the residual quad of
// this row (prefetched one row ahead).  Same arithmetic, same rounding points.
template <int EPI, bool F16 = false>          // F16: bias, residual and result are IEEE half instead of bfloat16 (the fp16 vision tower); same rounding points
__device__ __forceinline__ s16x4_t big_value_pre(const f32x4_t& a, bool has_scale, const f32x4_t& sc, bool has_bias, const s16x4_t& b, const s16x4_t& rr) {
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (has_scale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= sc[r];
    }
    if (has_bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += raw2f<F16>((uint16_t)b[r]);
    }
    if constexpr (EPI == EPI_GELU_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(raw2f<F16>(f2raw<F16>(v[r])));
    } else if constexpr (EPI == EPI_GELU_ERF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(raw2f<F16>(f2raw<F16>(v[r])));
    } else if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = raw2f<F16>(f2raw<F16>(v[r])) + raw2f<F16>((uint16_t)rr[r]);
    }
    return s16x4_t{(short)f2raw<F16>(v[0]), (short)f2raw<F16>(v[1]), (short)f2raw<F16>(v[2]), (short)f2raw<F16>(v[3])};
}
template <int EPI>
__device__ __forceinline__ void big_store(const GemmP& p, int m, int n, const f32x4_t& a) {
    if (m >= p.M) return;
    *reinterpret_cast<s16x4_t*>((bf16_t*)p.Y + (long long)m * p.ldy + n) = big_value<EPI>(p, m, n, a);
}
// n_gate = weight row of the gate quad (its up partner sits 16 rows further); ws = per-row weight scales or null
__device__ __forceinline__ s16x4_t big_value_swiglu(const f32x4_t& g_in, const f32x4_t& u_in, const float* ws = nullptr, int n_gate = 0) {
    f32x4_t g = g_in, u = u_in;
    if (ws) { g *= *reinterpret_cast<const f32x4_t*>(ws + n_gate); u *= *reinterpret_cast<const f32x4_t*>(ws + n_gate + 16); }
    s16x4_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float gg = bf2f(f2bf(g[r])), uu = bf2f(f2bf(u[r]));
        float sl = bf2f(f2bf(gg * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-gg * 1.4426950408889634f))));
        o[r] = (short)f2bf(sl * uu);
    }
    return o;
}
// Two neighbouring 16-column tiles, each lane holding 4 columns (lq*4..) of both: one v_permlane16_swap per dword turns that into
// 8 consecutive columns of ONE tile per lane (tile lq&1, columns (lq>>1)*8..), i.e. a 16-byte store per lane and 64 contiguous
// bytes per output row and instruction instead of 2 x 32.
__device__ __forceinline__ s16x8_t pair_to_row8(const s16x4_t& ta, const s16x4_t& tb) {
    const uint2 a = __builtin_bit_cast(uint2, ta), b = __builtin_bit_cast(uint2, tb);
    const auto x = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const auto y = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    const uint4 o = {x[0], y[0], x[1], y[1]};
    return __builtin_bit_cast(s16x8_t, o);
}
__device__ __forceinline__ void big_store_swiglu(const GemmP& p, int m, int n_gate, const f32x4_t& g, const f32x4_t& u) {
    if (m >= p.M) return;
    const int oc = (n_gate >> 5) * 16 + (n_gate & 15);
    *reinterpret_cast<s16x4_t*>((bf16_t*)p.Y + (long long)m * p.ldy + oc) = big_value_swiglu(g, u, p.wscale, n_gate);
}


// ------------------------------------------------------------------------------------------------------------------
// gemm_ringx_kernel<EPI, WN, M32, NS, EARLY>: the ring GEMM -- the MFMA-bound kernel of the path (every tower / projector GEMM of a
// 35-frame batch, gate_up and, with split-K over grid.z, down_proj of a >= 600-row LLM chunk).
//
// Block tile 256 x 64*WN, wave tile 128 x 64, K in BK = 32 slices through an NS-slot ring filled by direct-to-LDS DMA
// (global_load_lds, 1 KB per wave instruction).  Every wave is software-pipelined against itself: in step s it issues the MFMAs of
// slice s from registers while (a) the fragments of slice s+1 arrive from LDS -- B into the other register set, A in place (row i of
// A is dead once its MFMAs are issued; the last rows travel in spare registers so the last LDS read is two rows old at the barrier)
// -- and (b) its DMAs of slice s+NS refill the slot slice s just vacated.  Rows are fenced with sched_barrier so the compiler cannot
// hoist a load above MFMAs (its waitcnt pass would drain it).  ONE counted `s_waitcnt vmcnt((NS-2)*NDMA) lgkmcnt(0)` + raw s_barrier
// per slice.  X pieces are 16 rows x 64 B with the 16-byte chunk index XOR-ed by g[(row>>2)&3], g = {0,3,2,1} (conflict-free b128
// fragment reads, SQ_LDS_BANK_CONFLICT = 0), W pieces are the packed MFMA fragment tiles.  The kernel is persistent (one block per
// resident slot looping over its tiles: the next tile's first slices are in flight while the current tile is converted and stored) and
// the epilogue exchanges neighbouring column groups across lanes (v_permlane16/32_swap) so every lane stores 16 contiguous bytes.
// Tile order: bijective XCD remap, then bands of up to 8 m-tiles swept m-fastest (for N > M this is the W-panel-stationary order).
//
// Piece-major activations (round 5; GemmP::x_pm / y_pm): an intermediate with ONE producer and ONE consumer that are both this kernel (the tower's fc1 + GELU -> fc2,
// gate_up + SwiGLU -> down of a chunk) is stored as [M / 16][N / 32] pieces of 16 rows x 32 elements, 1 KB contiguous each, rows of 64 B inside.  The epilogue's wave
// instruction already writes exactly one such piece (16 rows x 64 B: 16 scattered row segments in the row-major form, one contiguous 1 KB burst here: the store probe of round 4
// measured 3.40 against 5.71 us per 33.5 MB), and the consumer's X piece is the same 1 KB run (16 row segments of 64 B in the row-major form).  Same values in other places.
//
// Shipped instantiation: WN = 4 (8 waves, 256 x 256), 16x16x32 MFMA, NS = 3, EARLY (the refill DMAs in the FIRST rows of a step).
// What the template parameters were built to test, on random operands, within one process (profiles/r02_gemm_shapes.json):
//   EARLY  refill DMAs issued right after the barrier instead of in the last rows: +3-5 % (1.20 -> 1.25 PF on gate_up at M = 1274).
//   NS = 4 one more slice of DMA lookahead (128 KB ring): +-0 -- DMA latency is covered at NS = 3.
//   M32    v_mfma_f32_32x32x16_bf16 (half the MFMA instructions, same LDS images: a 32-row operand = two neighbouring 16-row pieces):
//          5-10 % SLOWER on every shape although its loop is as clean in the ISA; not shipped.
//   WN = 2 4-wave 256 x 128 blocks, two unsynchronised blocks per CU (72 KB rings), optionally started half a tile apart so that one
//          block's tile seam is covered by the other's MFMAs: loses 10-15 % on long-K shapes (1.5 x the DMA bytes per flop), +-0 on the
//          K = 1152 tower shapes with or without the stagger; it wins only where 256 x 256 tiles cannot fill the chip.
//   DBG    timing-only modes (wrong results): with NO DMA at all the same loop runs 1.15-1.46 PF, with the real DMA 0.9-1.25 PF: the
//          ceiling of this structure is the wave-level issue stream (32 MFMA + 12 ds_read_b128 + 4 LDS-DMA per wave and slice, two
//          waves per SIMD), not HBM / L2 (FETCH_SIZE 0.6 GB per gate_up launch = 2 TB/s) and not LDS conflicts (0).
//          rocprofv3 PMC on gate_up: SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM cycles) = 56-60 %; per wave 23 % issuing, 47 % waiting
//          for the matrix pipe, 30 % parked at the slice barrier (profiles/r02_pmc_gemm.txt).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4_t quad_of(const f32x16_t& v, int q) { return f32x4_t{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; }
// two neighbouring n-quads (4 bf16 each) held by the two half-waves -> 8 consecutive columns per lane (v_permlane32_swap):
// lanes 0-31 end up with columns 0..7 of the 16-column group, lanes 32-63 with columns 8..15
__device__ __forceinline__ s16x8_t halves_to_row8(const s16x4_t& q0, const s16x4_t& q1) {
    const uint2 a = __builtin_bit_cast(uint2, q0), b = __builtin_bit_cast(uint2, q1);
    const auto x = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    const uint4 o = {x[0], y[0], x[1], y[1]};
    return __builtin_bit_cast(s16x8_t, o);
}

// DBG != 0: timing experiments only (results are WRONG): 1 = X pieces fetched as contiguous 1 KB runs, 2 = no X DMA, 3 = no DMA at all
template <int EPI, int WN, bool M32, int NS, bool EARLY, int DBG = 0, bool F16 = false>          // F16: IEEE-half operands / bias / residual / output (v_mfma_f32_16x16x32_f16: same rate, same layouts)
__global__ __launch_bounds__(WN * 128, 2) void gemm_ringx_kernel(GemmP p, int KT) {
    static_assert(!F16 || (!M32 && EPI != EPI_SWIGLU), "the fp16 form exists for the 16x16x32 schedule and the tower's epilogues");
    constexpr int BM = 256, BN = 64 * WN, BK = 32, NW = 2 * WN;
    constexpr int XE = BM * BK, WE = BN * BK, SE = XE + WE;    // elements per ring slot (32 KB / 24 KB)
    constexpr int XP = (BM / 16) / NW, WP = (BN / 16) / NW;    // X / W pieces (1 KB DMAs) per wave and slice: 2+2 (8 waves), 4+2 (4 waves)
    constexpr int NDMA = DBG == 3 ? 0 : (DBG == 2 ? WP : XP + WP);     // DMAs really issued per wave and slice (what vmcnt counts)
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // the wave index as a SCALAR: LDS-DMA destinations (M0) and tile offsets stay out of the VGPRs
    const int lr = lane & 15, lq = lane >> 4, lh = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;
    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM;
    const int nblk = nbx * nby;
    int m0 = 0, n0 = 0;
    auto tile_origin = [&](int bid) {
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int TB = nby < 8 ? nby : 8;                              // band height in m-tiles
        const int band = bid / (TB * nbx), rem = bid - band * (TB * nbx);
        const int tb = min(TB, nby - band * TB);                       // the last band may be shorter
        const int nt = rem / tb, mt = band * TB + rem - nt * tb;
        m0 = mt * BM; n0 = nt * BN;
    };
    const int ntiles = p.N >> 4;
    const bf16_t* X = (const bf16_t*)p.X;
    const bf16_t* Wp = (const bf16_t*)p.W;
    const int nsteps_all = p.K / BK;
    const int zsteps = (nsteps_all + gridDim.z - 1) / gridDim.z;
    const int t0 = blockIdx.z * zsteps;
    const int nsteps = min(nsteps_all, t0 + zsteps) - t0;

    const int srow = lane >> 2, spos = lane & 3;
    const int sswz = (0x1230 >> (((srow >> 2) & 3) * 4)) & 3;
    // DMA sources as UNIFORM base (scalar: tile, K step) + per-lane 32-bit byte offset (the saddr form of global_load_lds: no 64-bit VALU address math per
    // DMA, one VGPR per X piece and none per W piece -- a W piece is 1 KB contiguous, lane l reads bytes 16 l..).  Operand sizes under 4 GB are a dispatch condition.
    unsigned xo[XP]; long long wo[WP];
    const unsigned wlane = lane * 16;
    const int xstep = p.x_pm ? 1024 : BK * 2;          // bytes from one K slice of an X piece to the next
    auto tile_sources = [&]() {
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int pi = wave + NW * j;
            if (p.x_pm) {          // piece-major X: the whole piece is ONE contiguous 1 KB run (row group clamped to the last one; its rows past M are never-written memory that only feeds masked output rows)
                int grp = (m0 >> 4) + pi; const int gmax = (p.M - 1) >> 4; grp = grp < gmax ? grp : gmax;
                xo[j] = (unsigned)(grp * (p.K >> 5)) * 1024u + (unsigned)(srow * 64 + ((spos ^ sswz) * 16));
                continue;
            }
            int row = m0 + pi * 16 + srow; row = row < p.M ? row : p.M - 1;
            xo[j] = (unsigned)(row * (int)p.ldx + ((spos ^ sswz) * 8)) * 2u;
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int pi = wave + NW * j;
            int ntile = n0 / 16 + pi; ntile = ntile < ntiles ? ntile : ntiles - 1;
            wo[j] = ((long long)ntile * KT * 512 + (long long)t0 * 512) * 2;          // uniform byte offset of the piece's first k-step
        }
    };
    auto dma_x = [&](int slot, int step, int j) {
        if constexpr (DBG == 2 || DBG == 3) return;
        if constexpr (DBG == 1) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (long long)min(m0 + (wave + NW * j) * 16, p.M - 16) * p.ldx + (long long)step * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void*)(lds + slot * SE + (wave + NW * j) * 512), 16, 0, 0);
            return;
        }
        const char* ub = (const char*)X + (long long)(t0 + step) * xstep;
        asm volatile("" : "+s"(ub));          // keep base + zext(offset) as written (hipcc would hoist X + offset into a 64-bit VGPR pair per piece and add the K step there)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + xo[j]),
                                         (__attribute__((address_space(3))) void*)(lds + slot * SE + (wave + NW * j) * 512), 16, 0, 0);
    };
    auto dma_w = [&](int slot, int step, int j) {
        if constexpr (DBG == 3) return;
        const char* ub = (const char*)Wp + wo[j] + (long long)step * 1024;
        asm volatile("" : "+s"(ub));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + wlane),
                                         (__attribute__((address_space(3))) void*)(lds + slot * SE + XE + (wave + NW * j) * 512), 16, 0, 0);
    };
    auto stage = [&](int slot, int step) {
#pragma unroll
        for (int j = 0; j < XP; ++j) dma_x(slot, step, j);
#pragma unroll
        for (int j = 0; j < WP; ++j) dma_w(slot, step, j);
    };
    // the k-th DMA of a slice, k in [0, NDMA): X pieces first
    auto dma_k = [&](int slot, int step, int k) { if (k < XP) dma_x(slot, step, k); else if (k < XP + WP) dma_w(slot, step, k - XP); };

    constexpr int VM_ONE = NDMA, VM_TWO = 2 * NDMA;       // vmcnt leaving one / two slices in flight
    constexpr int NST = EPI == EPI_SWIGLU ? 8 : 16;       // output store instructions per wave and tile (16x16x32 schedule), ALWAYS issued (masked lanes write to a dump slot)
#ifdef MMDUET_NO_PST
    constexpr bool PST = false;                            // A/B build (tools/probes/nt_ab.sh)
#else
    constexpr bool PST = !M32 && NS == 3 && DBG == 0;      // output stores may stay pending across the next tile's start
#endif
#define RINGX_WAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(n) : "memory")
    const int G = gridDim.x;
    const int rswz = (0x1230 >> (((lr >> 2) & 3) * 4)) & 3;
    if constexpr (WN == 2 && NS == 3) {
        // Two blocks share a CU and would run in lockstep (same start, same tile length): both in their tile seam at the same time.  The
        // blocks dispatched into the second slot (ids >= 256, observed placement -- speed only) start ~half a tile late, so that one
        // block's seam (drain, epilogue math, store burst) is covered by the other's MFMAs for the rest of the launch.
        if (G > 256 && (int)blockIdx.x >= 256 && p.kper > 0) {
            for (int i = 0; i < p.kper; ++i) __builtin_amdgcn_s_sleep(127);          // 127 x 64 cycles ~ 4 us each
        }
    }

    if constexpr (!M32) {
        // ---------------- 16x16x32: 8 x 4 accumulator tiles, the schedule of gemm_ring256_kernel ----------------
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
        const int aoff = (wr * 8) * 512 + lr * 32 + ((lq ^ rswz) * 8);
        const int boff = XE + (wc * 4) * 512 + lane * 8;
        bf16x8_t a[8], b0[4], b1[4];
        auto step = [&](auto steady, int s, int slot, const bf16x8_t (&b)[4], bf16x8_t (&bn)[4], auto pending) {
            constexpr bool STEADY = decltype(steady)::value;
            constexpr bool PEND = decltype(pending)::value;          // first two slices of a tile whose predecessor's NST stores may still be in flight (see the epilogue)
            if (DBG == 5 && s < 2) RINGX_WAIT((NS - 2) * NDMA + 16);          // timing experiment: leave the previous tile's 16 stores pending (WRONG on the first tile)
            else if (PEND) RINGX_WAIT((NS - 2) * NDMA + NST);
            else if (STEADY || s + NS - 1 < nsteps) RINGX_WAIT((NS - 2) * NDMA);
            else if (NS == 4 && s + 2 < nsteps) RINGX_WAIT(NDMA);
            else RINGX_WAIT(0);
            if (!(DBG == 8 && (s & 1))) MMD_BAR();          // DBG 8 (timing only, RACY): the barrier of every other slice left out -- what would a barrier per TWO slices buy?
            const bool refill = STEADY || s + NS < nsteps;
            const bool more = STEADY || s + 1 < nsteps;
            const bf16_t* nbase = lds + (slot == NS - 1 ? 0 : slot + 1) * SE;
            bf16x8_t a6n, a7n;
            __builtin_amdgcn_s_setprio(1);
            if constexpr (DBG == 7) {
                // experiment (correct results): every LDS read / DMA sits BETWEEN two MFMAs of a row instead of behind the row's four, so its issue hides in the
                // matrix pipe's shadow of this wave's own previous MFMA.  Row i refills a[i-1] (dead since row i-1); a[7] travels in a7n from row 0.
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i][0] = mfma16<F16>(b[0], a[i], acc[i][0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) { if (i == 0) a7n = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + 7 * 512); else a[i - 1] = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + (i - 1) * 512); }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][1] = mfma16<F16>(b[1], a[i], acc[i][1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && i < 4) bn[i] = *reinterpret_cast<const bf16x8_t*>(nbase + boff + i * 512);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][2] = mfma16<F16>(b[2], a[i], acc[i][2]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (refill && i < XP + WP) dma_k(slot, s + NS, i);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i][3] = mfma16<F16>(b[3], a[i], acc[i][3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more) a[7] = a7n;
                __builtin_amdgcn_s_setprio(0);
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(b[j], a[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (i < 6) a[i] = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + i * 512);
                    if (i < 4) bn[i] = *reinterpret_cast<const bf16x8_t*>(nbase + boff + i * 512);
                    if (i == 0) a6n = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + 6 * 512);
                    if (i == 1) a7n = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + 7 * 512);
                }
                if (refill) {       // NDMA = 4: rows 4..7; NDMA = 6: rows 2..7
                    if (EARLY ? i < XP + WP : i >= 8 - (XP + WP)) dma_k(slot, s + NS, i - (EARLY ? 0 : 8 - (XP + WP)));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) { a[6] = a6n; a[7] = a7n; }
            __builtin_amdgcn_s_setprio(0);
        };
        auto prologue = [&]() { stage(0, 0); if (nsteps > 1) stage(1, 1); if (nsteps > 2) stage(2, 2); if (NS > 3 && nsteps > 3) stage(3, 3); };
        int tile = blockIdx.x;
        tile_origin(tile); tile_sources(); prologue();
        for (; tile < nblk; tile += G) {
            // pend: this block's previous tile left exactly NST output stores in flight behind the DMAs of slices 0..2 issued before them.  vmcnt retires loads
            // and stores in issue order (hipcc itself waits vmcnt(2) for "load; store; store; use"), so slice 0 has landed once at most (its 2 NDMA younger DMAs +
            // NST stores) are outstanding -- the stores need not drain here, nor in front of slices 1 and 2 (step's PEND waits); slice 3's wait is behind them.
            const bool pend = PST && tile != (int)blockIdx.x && nsteps >= NS + 3 && gridDim.z == 1;
            if (DBG == 5) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_TWO + 16) : "memory");
            else if (pend) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_TWO + NST) : "memory");
            else if (NS > 3 && nsteps > 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(3 * NDMA) : "memory");
            else if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_TWO) : "memory");
            else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_ONE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MMD_BAR();
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(lds + aoff + i * 512);
#pragma unroll
            for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const bf16x8_t*>(lds + boff + j * 512);
            int slot = 0, s = 0;
            if (pend) {       // (pend implies nsteps >= NS + 3: the first pair is a steady pair)
                step(std::true_type{}, 0, 0, b0, b1, std::true_type{});
                step(std::true_type{}, 1, 1, b1, b0, std::true_type{});
                slot = 2 % NS; s = 2;
            }
            for (; s + NS + 1 < nsteps; s += 2) {
                step(std::true_type{}, s, slot, b0, b1, std::false_type{});
                slot = slot == NS - 1 ? 0 : slot + 1;
                step(std::true_type{}, s + 1, slot, b1, b0, std::false_type{});
                slot = slot == NS - 1 ? 0 : slot + 1;
            }
            for (; s < nsteps; s += 2) {
                step(std::false_type{}, s, slot, b0, b1, std::false_type{});
                slot = slot == NS - 1 ? 0 : slot + 1;
                if (s + 1 < nsteps) {
                    step(std::false_type{}, s + 1, slot, b1, b0, std::false_type{});
                    slot = slot == NS - 1 ? 0 : slot + 1;
                }
            }
            const int em0 = m0, en0 = n0;
            bf16_t* const dump = (bf16_t*)p.dump + tid * 8;        // masked lanes store here: the store COUNT per wave must not depend on the tile (16 bytes per thread, content irrelevant)
            if (tile + G < nblk) { tile_origin(tile + G); tile_sources(); prologue(); }
            __builtin_amdgcn_sched_barrier(0);          // the output stores below stay BEHIND the next tile's first DMAs (the counted waits rely on that order)
            if (gridDim.z > 1) {
                float* wsl = p.ws + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = em0 + wr * 128 + i * 16 + lr;
                    if (m < p.M) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int nb = en0 + wc * 64 + j * 16; if (nb + 16 <= p.N) *reinterpret_cast<f32x4_t*>(wsl + (long long)m * p.N + nb + lq * 4) = acc[i][j]; }
                    }
                }
                return;
            }
            // per-column operands of this wave's four 16-column tiles: ONE load per tile (they do not depend on the row); the residual quads of a row are
            // fetched while the previous row is converted and stored -- a load issued BEFORE a row's stores is older than them, so waiting for it does not
            // drain them (the compiler, left alone, loaded one residual quad at a time with a full vmcnt(0) after each)
            f32x4_t scq[4]; s16x4_t biq[4], rq[4], rnext[4];
            int ncol[4];
            const bool has_sc = p.wscale != nullptr, has_bi = p.bias != nullptr;
            if constexpr (EPI != EPI_SWIGLU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nb = en0 + wc * 64 + (j & ~1) * 16;
                    ncol[j] = ((j & 1) ? (nb + 32 <= p.N ? nb + 16 : p.N - 16) : (nb + 16 <= p.N ? nb : p.N - 16)) + lq * 4;      // N tail: clamp the reads, mask the stores
                    scq[j] = has_sc ? *reinterpret_cast<const f32x4_t*>(p.wscale + ncol[j]) : f32x4_t{1, 1, 1, 1};
                    biq[j] = has_bi ? *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.bias + ncol[j]) : s16x4_t{0, 0, 0, 0};
                    rq[j] = rnext[j] = s16x4_t{0, 0, 0, 0};
                }
                if constexpr (EPI == EPI_RESID) {
                    const int m = min(em0 + wr * 128 + lr, p.M - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rnext[j] = *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.R + (long long)m * p.ldr + ncol[j]);
                }
            }
            if constexpr (DBG == 4) {          // timing only: no conversion, no stores (the accumulators stay live)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][1]), "v"(acc[i][j][2]), "v"(acc[i][j][3]));
            } else
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = em0 + wr * 128 + i * 16 + lr;
                const int mc = m < p.M ? m : p.M - 1;
                if constexpr (EPI == EPI_RESID) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rq[j] = rnext[j];
                    if (i < 7) {
                        const int mn = min(m + 16, p.M - 1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) rnext[j] = *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.R + (long long)mn * p.ldr + ncol[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);          // keep the next row's residual loads AHEAD of this row's stores (older => waiting for them leaves the stores in flight)
                }
                if constexpr (EPI == EPI_SWIGLU) {
                    const int nb = en0 + wc * 64;
                    const int ob = (nb >> 5) * 16;
                    const int ns0 = min(nb, p.N - 32) + lq * 4, ns1 = min(nb + 32, p.N - 32) + lq * 4;      // N tail: clamp the scale reads
                    const s16x8_t v = pair_to_row8(big_value_swiglu(acc[i][0], acc[i][1], p.wscale, ns0), big_value_swiglu(acc[i][2], acc[i][3], p.wscale, ns1));
                    const bool ok = m < p.M && nb + 32 * (lq & 1) + 32 <= p.N;
                    bf16_t* const yd = p.y_pm ? (bf16_t*)p.Y + ((long long)(m >> 4) * (p.N >> 6) + (ob >> 5)) * 512 + lr * 32 + (lq & 1) * 16 + (lq >> 1) * 8
                                              : (bf16_t*)p.Y + (long long)m * p.ldy + ob + (lq & 1) * 16 + (lq >> 1) * 8;          // piece-major: this wave instruction writes ONE contiguous 1 KB piece
                    if constexpr (PST) *reinterpret_cast<s16x8_t*>(ok ? yd : dump) = v;
                    else if (ok) *reinterpret_cast<s16x8_t*>(yd) = v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const int nb = en0 + wc * 64 + j * 16;
                        const s16x8_t v = pair_to_row8(big_value_pre<EPI, F16>(acc[i][j], has_sc, scq[j], has_bi, biq[j], rq[j]), big_value_pre<EPI, F16>(acc[i][j + 1], has_sc, scq[j + 1], has_bi, biq[j + 1], rq[j + 1]));
                        if constexpr (DBG == 6) { const u32x4_t w = __builtin_bit_cast(u32x4_t, v); asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3])); }      // timing only: converted, not stored
                        else {
                            const bool ok = m < p.M && nb + 16 * (lq & 1) + 16 <= p.N;
                            bf16_t* const yd = p.y_pm ? (bf16_t*)p.Y + ((long long)(m >> 4) * (p.N >> 5) + (nb >> 5)) * 512 + lr * 32 + (lq & 1) * 16 + (lq >> 1) * 8
                                                      : (bf16_t*)p.Y + (long long)m * p.ldy + nb + (lq & 1) * 16 + (lq >> 1) * 8;
                            if constexpr (PST) *reinterpret_cast<s16x8_t*>(ok ? yd : dump) = v;
                            else if (ok) *reinterpret_cast<s16x8_t*>(yd) = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
        }
    } else {
        // ---------------- 32x32x16: 4 (m) x 2 (n) accumulator tiles of 32 x 32 ----------------
        f32x16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int lt = (lane >> 4) & 1;                      // which 16-row piece of the 32-row operand this lane reads
        // X operand of m-tile i, k-step kk:  piece (wr*8 + 2i + lt), row lr, chunk (2kk + lh) ^ swz(lr)
        const int xoff0 = (wr * 8 + lt) * 512 + lr * 32 + (((0 + lh) ^ rswz) * 8);
        const int xoff1 = (wr * 8 + lt) * 512 + lr * 32 + (((2 + lh) ^ rswz) * 8);
        // W operand of n-tile j, k-step kk:  piece (wc*4 + 2j + lt), fragment slot (2kk + lh)*16 + lr
        const int woff0 = XE + (wc * 4 + lt) * 512 + ((0 + lh) * 16 + lr) * 8;
        const int woff1 = XE + (wc * 4 + lt) * 512 + ((2 + lh) * 16 + lr) * 8;
        bf16x8_t xf[4][2], w0[2][2], w1[2][2];
        auto ldx = [&](const bf16_t* base, int i, int kk) { return *reinterpret_cast<const bf16x8_t*>(base + (kk ? xoff1 : xoff0) + i * 1024); };
        auto ldw = [&](const bf16_t* base, int j, int kk) { return *reinterpret_cast<const bf16x8_t*>(base + (kk ? woff1 : woff0) + j * 1024); };
        auto step = [&](auto steady, int s, int slot, const bf16x8_t (&w)[2][2], bf16x8_t (&wn)[2][2]) {
            constexpr bool STEADY = decltype(steady)::value;
            if (STEADY || s + NS - 1 < nsteps) RINGX_WAIT((NS - 2) * NDMA);
            else if (NS == 4 && s + 2 < nsteps) RINGX_WAIT(NDMA);
            else RINGX_WAIT(0);
            MMD_BAR();
            const bool refill = STEADY || s + NS < nsteps;
            const bool more = STEADY || s + 1 < nsteps;
            const bf16_t* nbase = lds + (slot == NS - 1 ? 0 : slot + 1) * SE;
            bf16x8_t x3n0, x3n1;                 // m-tile 3 of the next slice travels in spare registers (issued in rows 0/1)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // row i: 4 MFMAs (two accumulators alternate, so a dependent pair is 64 cycles apart), then this row's share of the
                // traffic, fenced so the scheduler cannot hoist a load above the MFMAs
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[j][kk], xf[i][kk], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (i < 3) { xf[i][0] = ldx(nbase, i, 0); xf[i][1] = ldx(nbase, i, 1); }
                    if (i == 0) { x3n0 = ldx(nbase, 3, 0); wn[0][0] = ldw(nbase, 0, 0); }
                    if (i == 1) { x3n1 = ldx(nbase, 3, 1); wn[0][1] = ldw(nbase, 0, 1); }
                    if (i == 2) { wn[1][0] = ldw(nbase, 1, 0); wn[1][1] = ldw(nbase, 1, 1); }
                }
                if (refill) {       // NDMA = 4: rows 2, 3 two each; NDMA = 6: rows 1, 2, 3 two each
                    constexpr int first = EARLY ? 0 : 4 - (XP + WP) / 2;
                    if (i >= first && i < first + (XP + WP) / 2) { dma_k(slot, s + NS, 2 * (i - first)); dma_k(slot, s + NS, 2 * (i - first) + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) { xf[3][0] = x3n0; xf[3][1] = x3n1; }
            __builtin_amdgcn_s_setprio(0);
        };
        auto prologue = [&]() { stage(0, 0); if (nsteps > 1) stage(1, 1); if (nsteps > 2) stage(2, 2); if (NS > 3 && nsteps > 3) stage(3, 3); };
        int tile = blockIdx.x;
        tile_origin(tile); tile_sources(); prologue();
        for (; tile < nblk; tile += G) {
            if (NS > 3 && nsteps > 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(3 * NDMA) : "memory");
            else if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_TWO) : "memory");
            else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_ONE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MMD_BAR();
#pragma unroll
            for (int i = 0; i < 4; ++i) { xf[i][0] = ldx(lds, i, 0); xf[i][1] = ldx(lds, i, 1); }
#pragma unroll
            for (int j = 0; j < 2; ++j) { w0[j][0] = ldw(lds, j, 0); w0[j][1] = ldw(lds, j, 1); }
            int slot = 0, s = 0;
            for (; s + NS + 1 < nsteps; s += 2) {
                step(std::true_type{}, s, slot, w0, w1);
                slot = slot == NS - 1 ? 0 : slot + 1;
                step(std::true_type{}, s + 1, slot, w1, w0);
                slot = slot == NS - 1 ? 0 : slot + 1;
            }
            for (; s < nsteps; s += 2) {
                step(std::false_type{}, s, slot, w0, w1);
                slot = slot == NS - 1 ? 0 : slot + 1;
                if (s + 1 < nsteps) {
                    step(std::false_type{}, s + 1, slot, w1, w0);
                    slot = slot == NS - 1 ? 0 : slot + 1;
                }
            }
            const int em0 = m0, en0 = n0;
            if (tile + G < nblk) { tile_origin(tile + G); tile_sources(); prologue(); }
            // lane holds, for m = .. + (lane & 31), the n-quads 8q + 4*lh (q = 0..3) of every 32-wide n-tile
            if (gridDim.z > 1) {
                float* wsl = p.ws + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = em0 + wr * 128 + i * 32 + (lane & 31);
                    if (m < p.M) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int nb = en0 + wc * 64 + j * 32;
                            if (nb + 32 <= p.N) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4_t*>(wsl + (long long)m * p.N + nb + 8 * q + 4 * lh) = quad_of(acc[i][j], q);
                            }
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = em0 + wr * 128 + i * 32 + (lane & 31);
                const int mc = m < p.M ? m : p.M - 1;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nb = en0 + wc * 64 + j * 32;
                    const int nbc = nb + 32 <= p.N ? nb : p.N - 32;            // N tail: clamp the reads, mask the stores
                    if constexpr (EPI == EPI_SWIGLU) {
                        // rows 0..15 of the n-tile are gate, 16..31 up, of output columns (nb >> 5) * 16 ..+15
                        const s16x8_t v = halves_to_row8(big_value_swiglu(quad_of(acc[i][j], 0), quad_of(acc[i][j], 2), p.wscale, nbc + 4 * lh),
                                                         big_value_swiglu(quad_of(acc[i][j], 1), quad_of(acc[i][j], 3), p.wscale, nbc + 8 + 4 * lh));
                        if (m < p.M && nb + 32 <= p.N) *reinterpret_cast<s16x8_t*>((bf16_t*)p.Y + (long long)m * p.ldy + (nb >> 5) * 16 + lh * 8) = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const s16x8_t v = halves_to_row8(big_value<EPI>(p, mc, nbc + 8 * q + 4 * lh, quad_of(acc[i][j], q)),
                                                             big_value<EPI>(p, mc, nbc + 8 * (q + 1) + 4 * lh, quad_of(acc[i][j], q + 1)));
                            if (m < p.M && nb + 32 <= p.N) *reinterpret_cast<s16x8_t*>((bf16_t*)p.Y + (long long)m * p.ldy + nb + 8 * q + lh * 8) = v;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
#undef RINGX_WAIT
}
