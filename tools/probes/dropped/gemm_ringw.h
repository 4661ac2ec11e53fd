// gemm_ringw.h -- gemm_ringw_kernel: the ring GEMM with the W operand taken OFF the LDS path (VERDICT r03 "next round" item 2).
//
// gemm_ringx_kernel stages both operands through the LDS ring: per K slice (BK = 32) and wave 2 + 2 LDS-DMAs (X, W pieces), 8 + 4 ds_read_b128 (A, B
// fragments) and 32 MFMAs.  The weights are ALREADY fragment-major in HBM (pack_w16x32_kernel: piece `lane` of the 1 KiB tile (n-tile, k-step) is that
// lane's MFMA operand), so a wave can load its four W fragments of a slice with four plain `global_load_dwordx4` -- uniform base + lane * 16 -- straight
// into VGPRs, NB - 1 slices ahead, through NB register buffers.  Per slice and CU that removes a third of the fragment reads and half of the LDS-DMA
// writes; the price is every W byte fetched by both waves that share a column group (wr = 0 / 1): 32 KB instead of 16 KB of W per slice and CU from L2.
// The ring then holds X only (16 KB per slot).
//
// Load accounting (hand-counted: the loads are inline asm, hipcc neither sees nor waits for them).  Step s issues, in this order,
//     W(s + LA)  4 global loads (rows 0..3)          LA = NB - 1
//     X(s + NS)  XP = 2 LDS-DMAs (rows 4, 5)          into the slot slice s vacated (its fragments were read during step s - 1; every wave passed the barrier)
// and starts with ONE counted wait + barrier that must cover X(s + 1) -- read from LDS during this step -- and W(s) -- consumed now:
//     NS = LA + 1:  X(s + 1) is the younger of the two (issued in step s - LA behind W(s));  younger than it: (LA - 1) whole steps  ->  vmcnt((LA - 1) * (4 + XP))
//     NS > LA + 1:  W(s) is the younger one (first thing issued in step s - LA);            younger: X of that step + (LA - 1) steps ->  vmcnt(XP + (LA - 1) * (4 + XP))
// The prologue of a tile issues what the virtual steps -NS .. -1 would have: X(0) .. and W(0 .. LA - 1) in exactly that order, so the steady counts hold from step 0.
// vmcnt retires loads and stores in issue order.  The epilogue's NST output stores per wave (ALWAYS NST: masked lanes store to a dump slot, as in gemm_ringx_kernel)
// are issued BEHIND the next tile's prologue, so the tile-start wait and the waits of steps 0 .. LA - 1 -- which need prologue loads only -- allow for NST more
// operations in flight and the stores drain under the first slices of the next tile; step LA needs W(LA), issued behind the stores: its plain count drains them.
#pragma once
#include "gemm_ring.h"

template <int N, typename F> __device__ __forceinline__ void ringw_static_for(F&& f) {          // f(integral_constant<0>) ... f(integral_constant<N - 1>)
    if constexpr (N > 0) { ringw_static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

template <int EPI, int NS, int NB, bool F16 = false>
__global__ __launch_bounds__(512, 2) void gemm_ringw_kernel(GemmP p, int KT) {
    static_assert((NS == 3 && NB == 3) || (NS == 4 && NB == 2), "instantiated ring / buffer depths (4 slots + 3 buffers measured no better than 3 + 3 and spills)");
    static_assert(!F16 || EPI != EPI_SWIGLU, "the fp16 form exists for the tower's epilogues");
    constexpr int BM = 256, BN = 256, BK = 32, NW = 8, WN = 4;
    constexpr int XE = BM * BK;                                 // elements per ring slot: X only (16 KB)
    constexpr int XP = 2;                                       // X pieces (1 KB DMAs) per wave and slice
    constexpr int LA = NB - 1;                                  // W lookahead in slices
    constexpr int PER = NS == NB ? NS : 4;          // steps after which slot and buffer indices repeat
    constexpr int ISSUE = 4 + XP;                               // vmcnt events a steady step issues
    constexpr int VM_STEP = (NS == LA + 1) ? (LA - 1) * ISSUE : XP + (LA - 1) * ISSUE;
    constexpr int NST = EPI == EPI_SWIGLU ? 8 : 16;              // output store instructions per wave and tile, ALWAYS issued
    constexpr int VM_START = 4 * LA + XP * (NS - 1);            // operations a full prologue issues behind X(0)
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave / WN, wc = wave % WN;
    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM;
    const int nblk = nbx * nby;
    int m0 = 0, n0 = 0;
    auto tile_origin = [&](int bid) {
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int TB = nby < 8 ? nby : 8;
        const int band = bid / (TB * nbx), rem = bid - band * (TB * nbx);
        const int tb = min(TB, nby - band * TB);
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
    unsigned xo[XP]; long long wo[4];
    const unsigned wlane = lane * 16;
    auto tile_sources = [&]() {
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int pi = wave + NW * j;
            int row = m0 + pi * 16 + srow; row = row < p.M ? row : p.M - 1;
            xo[j] = (unsigned)(row * (int)p.ldx + ((spos ^ sswz) * 8)) * 2u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ntile = n0 / 16 + wc * 4 + j; ntile = ntile < ntiles ? ntile : ntiles - 1;
            wo[j] = ((long long)ntile * KT * 512 + (long long)t0 * 512) * 2;
        }
    };
    auto dma_x = [&](int slot, int step, int j) {
        const char* ub = (const char*)X + (long long)(t0 + step) * (BK * 2);
        asm volatile("" : "+s"(ub));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + xo[j]),
                                         (__attribute__((address_space(3))) void*)(lds + slot * XE + (wave + NW * j) * 512), 16, 0, 0);
    };
    // one W fragment (n-tile j of this wave, K slice `step`) into registers: uniform 64-bit base + lane * 16.  Inline asm: the destination is unprotected until
    // the counted wait of the step that consumes it (cdna_hip_programming.md 5.7 item 1)
    auto load_w = [&](bf16x8_t& dst, int j, int step) {
        const char* ub = (const char*)Wp + wo[j] + (long long)step * 1024;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(wlane), "s"(ub) : "memory");
    };
#define RINGW_WAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(n) : "memory")
    const int G = gridDim.x;
    const int rswz = (0x1230 >> (((lr >> 2) & 3) * 4)) & 3;
    bf16_t* const tbuf = lds + NS * XE + wave * 1024;         // this wave's two 1 KB transposition slots, behind the ring
    const bool direct = (p.flags & 1) != 0;
    if (p.kper > 0 && gridDim.z == 1 && (nblk - (int)blockIdx.x + G - 1) / G < (nblk + G - 1) / G) {          // free slack: this block owns one tile fewer
        for (int i = 0; i < p.kper; ++i) __builtin_amdgcn_s_sleep(127);                                      // 127 x 64 cycles ~ 4 us each
    }
    // one output store instruction: v = this lane's 16 bytes (row `lr` of the 16-row group, piece c = (lq & 1) * 2 + (lq >> 1) of the 64-byte span at element column
    // `col0`; `tw` = elements per 16-byte... per half span that must lie inside N for the piece to be stored).  Lane-adjacent form: row = lane >> 2, piece = lane & 3.
    auto store_span = [&](const s16x8_t& v, int mrow0, int col0, int tile_w, bf16_t* dump_at, const s16x8_t* resid, int slot = 0) {
        if (direct && !resid) {
            const int m = mrow0 + lr;
            const bool ok = m < p.M && col0 + tile_w * (lq & 1) + tile_w <= p.N;
            *reinterpret_cast<s16x8_t*>(ok ? (bf16_t*)p.Y + (long long)m * p.ldy + (tile_w == 16 ? col0 : (col0 >> 5) * 16) + (lq & 1) * 16 + (lq >> 1) * 8 : dump_at) = v;
            return;
        }
        *reinterpret_cast<s16x8_t*>(tbuf + slot * 512 + (lr * 4 + (lq & 1) * 2 + (lq >> 1)) * 8) = v;
        s16x8_t t = *reinterpret_cast<const s16x8_t*>(tbuf + slot * 512 + lane * 8);
        if (resid) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (short)f2raw<F16>(raw2f<F16>((uint16_t)t[e]) + raw2f<F16>((uint16_t)(*resid)[e]));
        }
        const int m = mrow0 + (lane >> 2), c = lane & 3;
        const bool ok = m < p.M && col0 + tile_w * (c >> 1) + tile_w <= p.N;
        *reinterpret_cast<s16x8_t*>(ok ? (bf16_t*)p.Y + (long long)m * p.ldy + (tile_w == 16 ? col0 : (col0 >> 5) * 16) + c * 8 : dump_at) = t;
    };

    // the two 64-byte spans (element columns col0, col0 + 32) of a 16-row group: both through the transposition, then both stored
    auto store_span2 = [&](const s16x8_t& v0, const s16x8_t& v1, int mrow0, int col0, bf16_t* dump_at) {
        if (direct) { store_span(v0, mrow0, col0, 16, dump_at, nullptr); store_span(v1, mrow0, col0 + 32, 16, dump_at, nullptr); return; }
        const int wslot = (lr * 4 + (lq & 1) * 2 + (lq >> 1)) * 8;
        *reinterpret_cast<s16x8_t*>(tbuf + wslot) = v0;
        *reinterpret_cast<s16x8_t*>(tbuf + 512 + wslot) = v1;
        const s16x8_t t0 = *reinterpret_cast<const s16x8_t*>(tbuf + lane * 8);
        const s16x8_t t1 = *reinterpret_cast<const s16x8_t*>(tbuf + 512 + lane * 8);
        const int m = mrow0 + (lane >> 2), c = lane & 3;
        bf16_t* const row = (bf16_t*)p.Y + (long long)m * p.ldy + col0 + c * 8;
        const bool ok0 = m < p.M && col0 + 16 * (c >> 1) + 16 <= p.N, ok1 = m < p.M && col0 + 32 + 16 * (c >> 1) + 16 <= p.N;
        *reinterpret_cast<s16x8_t*>(ok0 ? row : dump_at) = t0;
        *reinterpret_cast<s16x8_t*>(ok1 ? row + 32 : dump_at) = t1;
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
    const int aoff = (wr * 8) * 512 + lr * 32 + ((lq ^ rswz) * 8);
    bf16x8_t a[8], wb[NB][4];

    // step s of a tile: MFMAs of slice s from a[] / wb[s % NB]; the X fragments of slice s + 1 arrive in place (row i of a[] is dead once its MFMAs are issued);
    // W(s + LA) -> wb[(s + LA) % NB] (the buffer slice s - 1 just left); X(s + NS) -> slot s % NS
    auto step = [&](auto steady, int s, auto slot_c, auto buf_c, int allow, auto pending) {
        constexpr bool STEADY = decltype(steady)::value;
        constexpr bool PEND = decltype(pending)::value;          // the previous tile's NST stores may still be in flight behind the loads this step needs
        constexpr int slot = decltype(slot_c)::value, cur = decltype(buf_c)::value, nxt = (cur + LA) % NB, nslot = (slot + 1) % NS;
        if (PEND) RINGW_WAIT(VM_STEP + NST);
        else if (STEADY) RINGW_WAIT(VM_STEP);
        else {          // tile tail: fewer loads were issued behind the ones this step needs (see the table at the call site)
            if (allow >= VM_STEP) RINGW_WAIT(VM_STEP);
            else if (allow >= ISSUE) RINGW_WAIT(ISSUE);
            else RINGW_WAIT(0);
        }
        MMD_BAR();
        __builtin_amdgcn_sched_barrier(0);          // no MFMA of this step above the wait: wb[cur] is an asm destination the compiler believes long defined
        const bool wmore = STEADY || s + LA < nsteps;
        const bool refill = STEADY || s + NS < nsteps;
        const bool more = STEADY || s + 1 < nsteps;
        const bf16_t* nbase = lds + nslot * XE;
        bf16x8_t a6n, a7n;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(wb[cur][j], a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (wmore && i < 4) load_w(wb[nxt][i], i, s + LA);
            if (more) {
                if (i < 6) a[i] = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + i * 512);
                if (i == 0) a6n = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + 6 * 512);
                if (i == 1) a7n = *reinterpret_cast<const bf16x8_t*>(nbase + aoff + 7 * 512);
            }
            if (refill && i >= 4 && i < 4 + XP) dma_x(slot, s + NS, i - 4);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) { a[6] = a6n; a[7] = a7n; }
        __builtin_amdgcn_s_setprio(0);
    };
    // what the virtual steps -NS .. -1 would have issued, in their order
    auto prologue = [&]() {
#pragma unroll
        for (int t = -NS; t < 0; ++t) {
            if (t + LA >= 0 && t + LA < nsteps) {
#pragma unroll
                for (int j = 0; j < 4; ++j) load_w(wb[(t + LA) % NB][j], j, t + LA);
            }
            if (t + NS < nsteps) {
#pragma unroll
                for (int j = 0; j < XP; ++j) dma_x((t + NS) % NS, t + NS, j);
            }
        }
    };
    // operations the prologue issues behind X(0): everything but X(0) itself
    auto prologue_younger = [&]() { int n = 0; for (int t = -NS; t < 0; ++t) { if (t + LA >= 0 && t + LA < nsteps) n += 4; if (t + NS < nsteps && t > -NS) n += XP; } return n; };

    int tile = blockIdx.x;
    tile_origin(tile); tile_sources(); prologue();
    for (; tile < nblk; tile += G) {
        // X(0) has landed once at most the operations issued behind it are outstanding (the previous tile's output stores are younger still: a count that does
        // not allow for them drains the ones it must -- conservative and correct)
        const bool pend = tile != (int)blockIdx.x && nsteps >= PER + NS && gridDim.z == 1;
        if (pend) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_START + NST) : "memory");
        else if (prologue_younger() >= VM_START) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_START) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MMD_BAR();
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(lds + aoff + i * 512);
        int s = 0;
        if (pend) {          // first period of a tile whose predecessor left its stores in flight (nsteps >= PER + NS: all of it steady)
            ringw_static_for<PER>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                step(std::true_type{}, u, std::integral_constant<int, u % NS>{}, std::integral_constant<int, u % NB>{}, 0, std::integral_constant<bool, (u < LA)>{});
            });
            s = PER;
        }
        // steady periods: every step of the period issues its full set and finds the full set behind the loads it needs
        for (; s + PER + NS <= nsteps; s += PER) {
            ringw_static_for<PER>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                step(std::true_type{}, s + u, std::integral_constant<int, u % NS>{}, std::integral_constant<int, u % NB>{}, 0, std::false_type{});
            });
        }
        // tail (and short tiles): the same steps with run-time issue conditions.  Operations younger than the ones step s needs = what the steps s - LA + 1 .. s - 1
        // (and, for NS > LA + 1, the X refill of step s - LA) really issued
        for (; s < nsteps; ) {
            ringw_static_for<PER>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s < nsteps) {
                    int allow = 0;
                    if (NS > LA + 1 && s - LA + NS < nsteps) allow += XP;
                    for (int t = s - LA + 1; t < s; ++t) { if (t + LA < nsteps) allow += 4; if (t + NS < nsteps) allow += XP; }
                    step(std::false_type{}, s, std::integral_constant<int, u % NS>{}, std::integral_constant<int, u % NB>{}, allow, std::false_type{});
                    ++s;
                }
            });
        }
        const int em0 = m0, en0 = n0;
        bf16_t* const dump = (bf16_t*)p.dump + tid * 8;        // masked lanes store here: the store COUNT per wave must not depend on the tile
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
        // Epilogue operands by inline-asm loads issued AHEAD of the next tile's prologue and waited for with counted vmcnt: a compiler-visible load here is waited
        // for with vmcnt(0), which drains the prologue (DMAs and W fragments of the next tile) and every pending output store.  Per-column bias quads of the wave's
        // four 16-column tiles (accumulator layout: columns .. + lq * 4); the residual is added AFTER the lane transposition (one 16-byte load per store instruction,
        // in the store's own row / column layout): rnd(rnd(x W^T + b) + r) either way.
        s16x4_t biq[4];
        int ncol[4];
        const bool has_bi = p.bias != nullptr;
        if constexpr (EPI != EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nb = en0 + wc * 64 + (j & ~1) * 16;
                ncol[j] = ((j & 1) ? (nb + 32 <= p.N ? nb + 16 : p.N - 16) : (nb + 16 <= p.N ? nb : p.N - 16)) + lq * 4;
                // UNCONDITIONAL load (no bias: 8 harmless bytes of the dump slot, zeroed after the wait): an asm destination defined on one side of a branch gets a
                // phi copy right behind the asm statement -- a copy of a register the load has not written yet
                const bf16_t* bp = has_bi ? (const bf16_t*)p.bias + ncol[j] : (const bf16_t*)p.dump;
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(biq[j]) : "v"(bp) : "memory");
            }
        }
        // residual piece of store k (k = 2 i + pair): row em0 + wr * 128 + i * 16 + (lane >> 2), columns en0 + wc * 64 + pair * 32 + (lane & 3) * 8 (clamped: masked lanes never store)
        s16x8_t rr[2];
        auto load_resid = [&](s16x8_t& dst, int k) {
            const int m = min(em0 + wr * 128 + (k >> 1) * 16 + (lane >> 2), p.M - 1);
            const int c = min(en0 + wc * 64 + (k & 1) * 32 + (lane & 3) * 8, p.N - 8);
            const bf16_t* rp = (const bf16_t*)p.R + (long long)m * p.ldr + c;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(rp) : "memory");
        };
        if constexpr (EPI == EPI_RESID) load_resid(rr[0], 0);
        const bool full_pro = nsteps >= NS;          // the next tile's prologue issues VM_START + XP operations (same K range for every tile of this launch)
        if (tile + G < nblk) { tile_origin(tile + G); tile_sources(); prologue(); }
        __builtin_amdgcn_sched_barrier(0);          // the output stores below stay BEHIND the next tile's prologue (the counted waits rely on that order)
        if (tile + G < nblk && full_pro) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_START + XP) : "memory");          // bias (and residual 0) landed; the prologue stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EPI != EPI_SWIGLU) {
            if (!has_bi) {
#pragma unroll
                for (int j = 0; j < 4; ++j) biq[j] = s16x4_t{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (EPI == EPI_SWIGLU) {
                const int nb = en0 + wc * 64;
                const int ns0 = min(nb, p.N - 32) + lq * 4, ns1 = min(nb + 32, p.N - 32) + lq * 4;
                const s16x8_t v = pair_to_row8(big_value_swiglu(acc[i][0], acc[i][1], nullptr, ns0), big_value_swiglu(acc[i][2], acc[i][3], nullptr, ns1));
                store_span(v, em0 + wr * 128 + i * 16, nb, 32, dump, nullptr);          // (the SwiGLU span: 32 weight rows per 16 output columns)
            } else if constexpr (EPI == EPI_RESID) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int nb = en0 + wc * 64 + j * 16;
                    const int k = 2 * i + (j >> 1);
                    if (k + 1 < 16) load_resid(rr[(k + 1) & 1], k + 1);
                    const s16x8_t v = pair_to_row8(big_value_pre<EPI_NONE, F16>(acc[i][j], false, f32x4_t{1, 1, 1, 1}, has_bi, biq[j], s16x4_t{0, 0, 0, 0}),          // the residual joins behind the transposition
                                                   big_value_pre<EPI_NONE, F16>(acc[i][j + 1], false, f32x4_t{1, 1, 1, 1}, has_bi, biq[j + 1], s16x4_t{0, 0, 0, 0}));
                    // residual k was issued one store earlier (k = 0: ahead of the prologue, already waited for); younger than it: store k - 1 and residual k + 1
                    if (k > 0) { if (k + 1 < 16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
                    __builtin_amdgcn_sched_barrier(0);
                    store_span(v, em0 + wr * 128 + i * 16, nb, 16, dump, &rr[k & 1], 0);
                }
            } else {
                // both 64-byte spans of the row group through the transposition together: one LDS round trip per two stores
                const int nb = en0 + wc * 64;
                const s16x8_t v0 = pair_to_row8(big_value_pre<EPI, F16>(acc[i][0], false, f32x4_t{1, 1, 1, 1}, has_bi, biq[0], s16x4_t{0, 0, 0, 0}),
                                                big_value_pre<EPI, F16>(acc[i][1], false, f32x4_t{1, 1, 1, 1}, has_bi, biq[1], s16x4_t{0, 0, 0, 0}));
                const s16x8_t v1 = pair_to_row8(big_value_pre<EPI, F16>(acc[i][2], false, f32x4_t{1, 1, 1, 1}, has_bi, biq[2], s16x4_t{0, 0, 0, 0}),
                                                big_value_pre<EPI, F16>(acc[i][3], false, f32x4_t{1, 1, 1, 1}, has_bi, biq[3], s16x4_t{0, 0, 0, 0}));
                store_span2(v0, v1, em0 + wr * 128 + i * 16, nb, dump);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
    }
#undef RINGW_WAIT
}
