// DROPPED EXPERIMENT (round 4; not compiled into the library): measured 67-80 us where gemm_big_kernel<64> takes 52 us -- profiles/r04_gemm_experiments.md section 6, profiles/r04_mid_ab.json.
// Kept as the record of what was built; it included into gemm.hip behind variant 400 + TNW.
// gemm_mid_kernel<TNW, NS, EPI>: the MFMA-bound GEMM of a chunk's q/k/v and o_proj (M ~ 1.3 k rows, N = 4608 / 3584, K = 3584) -- round 4, VERDICT r03 item 2.
//
// What it replaces there: gemm_big_kernel<64> (128 x 64 tiles, two LDS buffers, a draining __syncthreads() per 64-deep K step).  Measured in round 3: 21 % MFMA busy,
// 59 % of the wave cycles parked at the barrier -- every K step waits for a DMA round trip that only other blocks' MFMA phases can cover -- 53 / 51 us against 33 / 26 us
// of MFMA time.  The 256 x 256 ring cannot help: 90 / 70 tiles for 256 CUs.
//
//   * block = 4 waves (2 x 2), tile 128 rows x (2 TNW 16-column tiles); the wave tile is 64 x 16 TNW.  TNW is chosen per shape so that ONE round of blocks fills the
//     256 CUs (M = 1274: qkv 10 x 24 blocks of 128 x 192, o_proj 10 x 23 of 128 x 160) -- launch_mid's cost model.
//   * K in 32-deep slices through an NS-slot LDS ring filled by LDS-DMA: X pieces in the ring GEMM's image (16 rows x 64 B, 16-byte chunks XOR-swizzled: conflict-free
//     b128 fragment reads), W pieces as they lie in HBM (fragment-major 1 KB tiles).  Every wave issues the SAME number of DMAs per slice (surplus ones re-load a piece).
//   * the wave is pipelined against itself: iteration s waits for slice s + 1 (ONE counted s_waitcnt vmcnt + raw s_barrier), issues the DMAs of slice s + NS - 1 into
//     the slot whose fragments were consumed an iteration ago, reads the fragments of s + 1 into the other register set and runs the MFMAs of s from the first.
//     NS - 2 slices stay in flight across every barrier.  One wave per SIMD (the tile's 4 TNW accumulator quads + two fragment sets), one block per CU.
//   * epilogue: per-column scale (fp8 weights) / bias, residual (in place), 16-byte row-contiguous stores through the lane exchange of the other tile kernels.
// Same values as gemm_big_kernel / the ring: products accumulate over K in slice order in fp32, one rounding at the end.
#pragma once

template <int TNW, int NS, int EPI>
__global__ __launch_bounds__(256) void gemm_mid_kernel(GemmP p, int KT, int nbm, int nbn) {
    constexpr int WPC = 2 * TNW, PIECES = 8 + WPC;
    constexpr int WJ = (WPC + 3) / 4;                        // W DMAs per wave and slice
    constexpr int PW = 2 + WJ;                               // DMAs per wave and slice
    constexpr int SLOT = PIECES * 512;                       // elements per ring slot
    constexpr int VM_STEADY = (NS - 3) * PW;
    static_assert(NS >= 3 && VM_STEADY <= 63, "ring depth");
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4, wy = wave >> 1, wx = wave & 1;
    int bid = blockIdx.x;
    {   // XCD x owns a contiguous run of tile ids (consecutive block ids go round-robin over the 8 XCDs)
        const int nblk = nbm * nbn, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int mt, nb;
    if (p.N > p.M) { nb = bid / nbm; mt = bid % nbm; } else { mt = bid / nbn; nb = bid % nbn; }          // the larger operand's panel is fetched once
    const int m0 = mt * 128, ntiles = p.N >> 4, nt0 = nb * WPC;
    const int nsl = KT;

    const int srow = lane >> 2, spos = lane & 3;
    const int sswz = (0x1230 >> (((srow >> 2) & 3) * 4)) & 3;
    const char* xsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int row = m0 + (wave + 4 * j) * 16 + srow; row = row < p.M ? row : p.M - 1;
        xsrc[j] = (const char*)p.X + ((long long)row * p.ldx + (spos ^ sswz) * 8) * 2;
    }
    const char* wsrc[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        int nt = nt0 + (wave + 4 * j) % WPC; nt = nt < ntiles ? nt : ntiles - 1;
        wsrc[j] = (const char*)p.W + (long long)nt * KT * 1024 + lane * 16;
    }
    auto issue = [&](int s, int slot) {
        bf16_t* base = lds + slot * SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[j] + (long long)s * 64),
                                             (__attribute__((address_space(3))) void*)(base + (wave + 4 * j) * 512), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < WJ; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + (long long)s * 1024),
                                             (__attribute__((address_space(3))) void*)(base + (8 + (wave + 4 * j) % WPC) * 512), 16, 0, 0);
    };
    const int rswz = (0x1230 >> (((lr >> 2) & 3) * 4)) & 3;
    const int aoff = lr * 32 + ((lq ^ rswz) * 8);
    auto frags = [&](int slot, bf16x8_t (&xf)[4], bf16x8_t (&wf)[TNW]) {
        const bf16_t* xs = lds + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(xs + (wy * 4 + i) * 512 + aoff);
#pragma unroll
        for (int j = 0; j < TNW; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(xs + (8 + wx * TNW + j) * 512 + lane * 8);
    };

    f32x4_t acc[4][TNW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
    bf16x8_t xa[4], wa[TNW], xb[4], wb[TNW];

    // prologue: slices 0 .. NS - 2 in flight, the fragments of slice 0 in set a
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) if (u < nsl) issue(u, u);
    if (nsl > NS - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NS - 2) * PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MMD_BAR();
    __builtin_amdgcn_sched_barrier(0);
    frags(0, xa, wa);

    int slot_next = 1 % NS;                                  // slot of slice s + 1
    int slot_fill = NS - 1;                                  // slot the DMAs of slice s + NS - 1 go to (= the slot of slice s - 1)
    auto step = [&](int s, bf16x8_t (&xc)[4], bf16x8_t (&wc)[TNW], bf16x8_t (&xn)[4], bf16x8_t (&wn)[TNW]) {
        if (s + 1 < nsl) {
            if (s + NS - 2 < nsl) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_STEADY) : "memory");          // slice s + 1 landed; the NS - 3 behind it stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             // tail: fewer were issued behind it
            MMD_BAR();
            __builtin_amdgcn_sched_barrier(0);
            if (s + NS - 1 < nsl) issue(s + NS - 1, slot_fill);
            frags(slot_next, xn, wn);
            slot_fill = slot_fill + 1 == NS ? 0 : slot_fill + 1;
            slot_next = slot_next + 1 == NS ? 0 : slot_next + 1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[j], xc[i], acc[i][j], 0, 0, 0);
    };
    for (int s = 0; s < nsl; s += 2) {
        step(s, xa, wa, xb, wb);
        if (s + 1 < nsl) step(s + 1, xb, wb, xa, wa);
    }

    // epilogue
    const bool has_sc = p.wscale != nullptr, has_bi = p.bias != nullptr;
    f32x4_t scq[TNW]; s16x4_t biq[TNW];
    const int ntw0 = nt0 + wx * TNW;
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        int nt = ntw0 + j; nt = nt < ntiles ? nt : ntiles - 1;
        const int n = nt * 16 + lq * 4;
        scq[j] = has_sc ? *reinterpret_cast<const f32x4_t*>(p.wscale + n) : f32x4_t{1, 1, 1, 1};
        biq[j] = has_bi ? *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.bias + n) : s16x4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wy * 64 + i * 16 + lr;
        const int mc = m < p.M ? m : p.M - 1;
        s16x4_t rq[TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            rq[j] = s16x4_t{0, 0, 0, 0};
            if constexpr (EPI == EPI_RESID) {
                int nt = ntw0 + j; nt = nt < ntiles ? nt : ntiles - 1;
                rq[j] = *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.R + (long long)mc * p.ldr + nt * 16 + lq * 4);
            }
        }
#pragma unroll
        for (int j = 0; j + 1 < TNW; j += 2) {
            const s16x8_t v = pair_to_row8(big_value_pre<EPI>(acc[i][j], has_sc, scq[j], has_bi, biq[j], rq[j]), big_value_pre<EPI>(acc[i][j + 1], has_sc, scq[j + 1], has_bi, biq[j + 1], rq[j + 1]));
            if (m < p.M && ntw0 + j + 1 < ntiles) *reinterpret_cast<s16x8_t*>((bf16_t*)p.Y + (long long)m * p.ldy + (ntw0 + j) * 16 + (lq & 1) * 16 + (lq >> 1) * 8) = v;
            else if (m < p.M && ntw0 + j < ntiles) {          // (a ragged last block whose pair is cut in half: the plain quad of the first tile)
                *reinterpret_cast<s16x4_t*>((bf16_t*)p.Y + (long long)m * p.ldy + (ntw0 + j) * 16 + lq * 4) = big_value_pre<EPI>(acc[i][j], has_sc, scq[j], has_bi, biq[j], rq[j]);
            }
        }
        if constexpr (TNW & 1) {
            constexpr int j = TNW - 1;
            if (m < p.M && ntw0 + j < ntiles)
                *reinterpret_cast<s16x4_t*>((bf16_t*)p.Y + (long long)m * p.ldy + (ntw0 + j) * 16 + lq * 4) = big_value_pre<EPI>(acc[i][j], has_sc, scq[j], has_bi, biq[j], rq[j]);
        }
    }
}

// Cost of a decomposition in units of one 16-column tile-slice per wave: rounds of blocks over the CUs x (wave tile width + a fixed part for the slice's
// barrier / DMA issue and the tile's prologue and epilogue).
struct MidPlan { int tnw, nbm, nbn, blocks; double cost; };
static inline MidPlan mid_plan(const GemmArgs& a, int force_tnw = 0) {
    MidPlan best{0, 0, 0, 0, 1e30};
    const int nbm = cdiv(a.M, 128), ntiles = a.N >> 4;
    for (int t = 3; t <= 9; ++t) {
        if (force_tnw && t != force_tnw) continue;
        const int nbn = cdiv(ntiles, 2 * t), blocks = nbm * nbn;
        const double cost = (double)cdiv(blocks, 256) * (t + 1.0);
        if (cost < best.cost) best = MidPlan{t, nbm, nbn, blocks, cost};
    }
    return best;
}
static inline bool mid_ok(int dtype, const GemmArgs& a) {
    return dtype == MMD_BF16 && !a.f16 && a.Wp != nullptr && a.M > 256 && (a.N % 16) == 0 && (a.K % 32) == 0 && (a.ldx % 8) == 0 && ((uintptr_t)a.X % 16) == 0 && !a.out_f32 &&
           (a.ldy % 8) == 0 && ((uintptr_t)a.Y % 16) == 0 && (a.epi == EPI_NONE || (a.epi == EPI_RESID && (a.ldr % 4) == 0 && ((uintptr_t)a.R % 8) == 0)) &&
           (a.bias == nullptr || ((uintptr_t)a.bias % 8) == 0) && !a.slabs_out && !a.chain;
}
template <int TNW, int NS = 5>
static hipError_t launch_mid_t(const GemmP& p, const GemmArgs& a, const MidPlan& pl, hipStream_t st) {
    const size_t smem = (size_t)NS * (8 + 2 * TNW) * 1024;
    const void* fn = a.epi == EPI_RESID ? (const void*)gemm_mid_kernel<TNW, NS, EPI_RESID> : (const void*)gemm_mid_kernel<TNW, NS, EPI_NONE>;
    if (smem > 65536) { hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); if (e != hipSuccess) return e; }
    if (a.epi == EPI_RESID) hipLaunchKernelGGL((gemm_mid_kernel<TNW, NS, EPI_RESID>), dim3(pl.blocks), dim3(256), smem, st, p, a.K >> 5, pl.nbm, pl.nbn);
    else hipLaunchKernelGGL((gemm_mid_kernel<TNW, NS, EPI_NONE>), dim3(pl.blocks), dim3(256), smem, st, p, a.K >> 5, pl.nbm, pl.nbn);
    return hipGetLastError();
}
static hipError_t launch_mid(const GemmP& p, const GemmArgs& a, hipStream_t st, int force = 0) {
    if (force >= 20) {          // 20 + 10 * (NS - 3) + TNW : small tiles, several blocks per CU
        const int ns = 3 + (force - 20) / 10, t = (force - 20) % 10;
        MidPlan q{t, cdiv(a.M, 128), cdiv(a.N >> 4, 2 * t), 0, 0}; q.blocks = q.nbm * q.nbn;
        if (a.plan_out) { a.plan_out[0] = GEMM_K_MID; a.plan_out[1] = q.blocks; a.plan_out[2] = 1; a.plan_out[3] = q.blocks; }
        if (t == 2 && ns == 3) return launch_mid_t<2, 3>(p, a, q, st);
        if (t == 2 && ns == 4) return launch_mid_t<2, 4>(p, a, q, st);
        if (t == 2 && ns == 5) return launch_mid_t<2, 5>(p, a, q, st);
        if (t == 3 && ns == 3) return launch_mid_t<3, 3>(p, a, q, st);
        if (t == 3 && ns == 4) return launch_mid_t<3, 4>(p, a, q, st);
        if (t == 4 && ns == 3) return launch_mid_t<4, 3>(p, a, q, st);
        return hipErrorInvalidValue;
    }
    const int force_tnw = force;
    const MidPlan pl = mid_plan(a, force_tnw);
    if (a.plan_out) { a.plan_out[0] = GEMM_K_MID; a.plan_out[1] = pl.nbm * pl.nbn; a.plan_out[2] = 1; a.plan_out[3] = pl.blocks; }
    switch (pl.tnw) {
        case 3: return launch_mid_t<3>(p, a, pl, st);
        case 4: return launch_mid_t<4>(p, a, pl, st);
        case 5: return launch_mid_t<5>(p, a, pl, st);
        case 6: return launch_mid_t<6>(p, a, pl, st);
        case 7: return launch_mid_t<7>(p, a, pl, st);
        case 8: return launch_mid_t<8>(p, a, pl, st);
        case 9: return launch_mid_t<9>(p, a, pl, st);
        default: return hipErrorInvalidValue;
    }
}
