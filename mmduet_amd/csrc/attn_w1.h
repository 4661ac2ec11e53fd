// attn_w1.h -- attn_gqa128_w1_kernel: the LLM attention (head_dim 128, bf16, causal GQA with a query offset over the KV arena; Qwen2Attention.forward,
// transformers qwen2/modeling_qwen2.py:200-240) with the matrix / vector overlap written INTO each wave's instruction stream.
// Included by attn.hip (uses its AttnP, xcd_block_id, quad_lanes_max, ATTN_DEFER).
//
// Why a second structure.  attn_gqa128_kernel<2, 4, 8> lets the hardware interleave one wave's MFMA segment with its SIMD partner's softmax segment (two barrier-separated
// segments per tile, the wave groups half a tile out of step).  Measured (profiles/r04_attention_status.md, r05_attn_small_s.md): the matrix pipe is busy a third of the
// time and 35-45 % of the wave cycles wait to issue -- each wave alternates between a stretch that only wants the matrix pipe and a stretch that only wants the VALU, and
// the pairing only works while both partners stay in step.  Here every wave runs a software pipeline over 32-key UNITS in which both kinds of work sit side by side:
//
//   unit u = one 32-key half tile.   A(u):  S(u+1) = K(u+1) Q^T   [8 RT MFMAs, 8 K fragment reads]   beside   P(u) = exp2(S(u) scale - m)                  [~25 RT VALU]
//                                    B(u):  O += V(u)^T P(u)      [8 RT MFMAs, 8 V fragment reads]   beside   l += sum P(u);  m, rescale factor of u + 1    [~22 RT VALU]
//
//   Both phases are single basic blocks (mask / rescale / barrier decisions sit between them); sched_group_barrier deals ~3 vector instructions into every MFMA's
//   16-cycle shadow.  Scores are double-buffered in registers (S(u+1) is produced while S(u) is consumed); the running maximum moves only when it grows by more than
//   2^ATTN_DEFER (as in attn_gqa128_kernel) and the rescale of O and l is applied between B(u) and A(u+1): after every product of unit u has been accumulated, before
//   any of unit u+1 (the order the deferred-maximum rule needs).  K / V tiles (64 keys: one contiguous 16 KB block of the arena each) arrive by LDS-DMA into a
//   four-slot ring, two tiles ahead of use, ONE barrier per tile, counted vmcnt; the swizzled LDS images and the fragment conventions are attn_gqa128_kernel's.
//   A block is 8 waves x up to RT = 2 row tiles of 16 stacked rows (row = tok * G + g); row tiles are dealt round-robin (tile k -> wave k % 8), and a wave runs the
//   program instantiated for the number of tiles it really holds -- the per-frame step (S = 49: 343 rows per kv head = 2 blocks of 11 tiles) wastes no matrix time on
//   padding.  <= 256 registers per wave: hipcc then emits the VGPR form of every MFMA (beyond 256 it switches ALL of them to the AGPR form and pays an accvgpr move per score).
//
// Rounding points = attn_gqa128_kernel's: scores fp32, statistics fp32, P rounded to bf16 before P.V, l = fp32 sum of the UNROUNDED probabilities.
#pragma once
#include <type_traits>
// timing-only ablations (WRONG results; tools/attn_w1_ablate.sh builds one debug library per value, never shipped): 1 no exp2, 2 no row-sum adds, 3 no score MFMAs,
// 4 no P.V MFMAs, 5 no fragment reads after the first three, 6 no tile barrier / DMA wait, 7 no maximum chain
#ifndef W1_DBG
#define W1_DBG 0
#endif

template <int RT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void attn_gqa128_w1_kernel(AttnP p) {
    constexpr int D = 128, KT = 64, TILE = KT * D, NSLOT = 4, NPC = 16 / WAVES;
    extern __shared__ __attribute__((aligned(16))) bf16_t kv[];        // NSLOT slots of (K tile, V^T tile): 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    int bx, by, bz; xcd_block_id(bx, by, bz);
    const int G = p.nh / p.nkv, kvh = by;
    const int rows_total = p.S * G;
    const int block_rows = p.block_rows;                                 // multiple of 16, <= 16 RT WAVES
    const int blk_first_row = bx * block_rows;
    const int blk_end_row = min(rows_total, blk_first_row + block_rows);
    const int blk_tiles = (blk_end_row - blk_first_row + 15) >> 4;
    const int my_tiles = __builtin_amdgcn_readfirstlane(wave < blk_tiles ? (blk_tiles - 1 - wave) / WAVES + 1 : 0);       // row tiles wave, wave + WAVES, ...
    const long long n_ctx = p.n_ctx, n_tot = n_ctx + p.S;
    const bf16_t* Kg = (const bf16_t*)p.K + kvh * p.k_hs;
    const bf16_t* Vg = (const bf16_t*)p.V + kvh * p.v_hs;

    // key range of this block / split (multiples of 64 except at the very end)
    const long long blk_limit = p.causal ? min(n_tot, n_ctx + (long long)((blk_end_row - 1) / G) + 1) : n_tot;
    const long long blk_min_limit = p.causal ? n_ctx + (long long)(blk_first_row / G) + 1 : n_tot;   // keys below this are visible to every row of the block
    const long long kbeg = (long long)bz * p.kv_per_split;
    const long long kend = min(blk_limit, kbeg + p.kv_per_split);
    const int ntile = kbeg < kend ? (int)((kend - kbeg + KT - 1) >> 6) : 0;

    // ---- staging (attn_gqa128_kernel's): per-lane source offsets fixed for the kernel, a scalar base per tile ----
    unsigned koff[NPC], voff[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int pc = wave + WAVES * j;
        const int key = pc * 4 + (lane >> 4);
        koff[j] = (unsigned)((key * D + (((lane & 15) ^ ((((key >> 3) & 3) << 2) | (key & 3))) * 8)) * 2);
        const int dim = pc * 8 + (lane >> 3);
        voff[j] = (unsigned)(((dim << 6) + (((lane & 7) ^ ((dim >> 1) & 7)) * 8)) * 2);
    }
    auto stage = [&](int t, int t_src = -1) {          // tile t_src (default t) into the slot of tile t
        bf16_t* ks = kv + (t & (NSLOT - 1)) * 2 * TILE;
        bf16_t* vt = ks + TILE;
        const long long k0 = kbeg + (long long)(t_src < 0 ? t : t_src) * KT;
        const char* kb = (const char*)(Kg + k0 * D);
        const char* vb = (const char*)(Vg + (((k0 >> 6) * D) << 6));
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int pc = wave + WAVES * j;
            unsigned ko = koff[j], vo = voff[j];
            asm volatile("" : "+v"(ko), "+v"(vo));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko), (__attribute__((address_space(3))) void*)(ks + pc * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vo), (__attribute__((address_space(3))) void*)(vt + pc * 512), 16, 0, 0);
        }
    };
    // tile t's K / V have landed for everybody and the slot of tile t - 2 (its V was last read two phases ago) is free: this wave's pieces of tile t are complete once at
    // most the pieces of tile t + 1 (2 NPC DMAs) are outstanding; then tile t + 2 goes out.
    auto sync_tile = [&](int t) {
        if constexpr (W1_DBG == 6) { if (t + 2 < ntile) stage(t + 2); return; }
        if (t + 1 < ntile || t + 1 <= 2) { if constexpr (NPC == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }          // (t + 1 <= 2: one of the three opening stages, real or not)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < ntile) stage(t + 2);
    };
    auto need_mask = [&](int tile) { const long long e = kbeg + (long long)(tile + 1) * KT; return e > blk_min_limit || e > kend; };
    const int vsw = (lr >> 1) & 7;
    const float scale = p.scale_log2;
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using MY = std::integral_constant<bool, true>; using MN = std::integral_constant<bool, false>;
    // ---- the wave program for RA row tiles (RA = 0: a wave without rows keeps staging and meeting the barriers) ----
    auto run = [&](auto RA_) {
        constexpr int RA = decltype(RA_)::value;
        constexpr int RB = RA > 0 ? RA : 1;
        stage(0, 0);
        int lim0[RB];
        bf16x8_t qf[RB][4];
#pragma unroll
        for (int rt = 0; rt < RA; ++rt) {
            const int row = blk_first_row + (rt * WAVES + wave) * 16 + lr;
            const bool ok = row < blk_end_row;
            const int tok = ok ? row / G : 0, head = kvh * G + (ok ? row % G : 0);
            const long long limit = !ok ? 0 : (p.causal ? n_ctx + tok + 1 : n_tot);                 // keys [0, limit) visible
            lim0[rt] = (int)((limit < kend ? limit : kend) - kbeg);                                           // (a split's key range fits 31 bits)
            const bf16_t* qrow = (const bf16_t*)p.q + (long long)tok * p.ldq + (long long)head * D;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const s16x8_t*>(qrow + c * 32 + lq * 8);
                qf[rt][c] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
        // hipcc cannot count LDS-DMAs: its wait for the q fragments (ordinary loads) is a vmcnt(0) wherever their first use lands.  Make that here, once, with only tile 0
        // in flight beside them (it is needed first anyway); tiles 1 and 2 go out behind it.  ALWAYS three stages (a split of fewer tiles re-reads its last tile into the
        // unused slots), so the hand-counted waits below see the same number of DMAs whatever the split holds
#pragma unroll
        for (int rt = 0; rt < RA; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" :: "v"(qf[rt][c]));
        stage(1, ntile > 0 ? min(1, ntile - 1) : 0);
        stage(2, ntile > 0 ? min(2, ntile - 1) : 0);

        f32x4_t oacc[RB][8];
        float m_run[RB], l_run[RB], neg_m[RB], alpha[RB];
#pragma unroll
        for (int rt = 0; rt < RA; ++rt) {
            m_run[rt] = -INFINITY; l_run[rt] = 0.f; neg_m[rt] = 0.f; alpha[rt] = 1.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) oacc[rt][t] = f32x4_t{0, 0, 0, 0};
        }
        f32x4_t st[2][RB][2];                // scores of two units in flight: [buffer][row tile][key sub-tile]
        unsigned pk[RB][4];                  // ... rounded to bf16, pairwise packed: the B operand of P.V
        float mxr[RB]; bool mvd[RB];          // phase B's vector chain, carried from step to step

        // The two phases are written STEP BY STEP, each step = the MFMAs of one fragment (one per row tile) + that step's share of the vector work + the LDS read of the
        // fragment two steps ahead, with sched_barrier(0) between steps: the order in the binary is the order written here (left alone hipcc clumps the MFMAs, hoists all
        // eight fragment reads -- 32 registers -- and spills; its sched_group_barrier solver did not find the interleave either).
        auto kfrag = [&](const bf16_t* Ks, int h, int g) {           // K fragment g = (c, t) = (g >> 1, g & 1) of half h
            const int c = g >> 1, t = g & 1;
            return *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * D + (((c * 4 + lq) ^ lr) * 8));
        };
        auto vfrag = [&](const bf16_t* Vt, int h, int t) {
            return *reinterpret_cast<const bf16x8_t*>(Vt + (t * 16 + lr) * KT + (((h * 4 + lq) ^ vsw) * 8));
        };
        // Fragments travel through ONE four-deep register ring shared by both phases, three reads ahead of their MFMAs; a phase's last three steps already read the
        // NEXT phase's first fragments (when that tile is known to have landed), so only the phase behind a tile barrier starts with an exposed LDS latency.
        bf16x8_t fr[4];
        // A: S(buffer BN) = K(tile, half h) Q^T  beside  P = exp2(S(buffer BC) scale - m) (rounded to bf16 for the matrix product, kept unrounded for the row sum).
        //    WITH_S = false: the last unit has no successor, only the vector half runs.  PRE: fragments 0..2 are already in the ring.  NEXT: read V(tile_v, hv)'s first three.
        auto phase_a = [&](auto BN_, auto WITH_S_, auto PRE_, auto NEXT_, int tile, int h, int tile_v, int hv) {
            constexpr int bn = decltype(BN_)::value, bc = 1 - bn;
            constexpr bool WITH_S = decltype(WITH_S_)::value, PRE = decltype(PRE_)::value, NEXT = decltype(NEXT_)::value;
            const bf16_t* Ks = kv + (tile & (NSLOT - 1)) * 2 * TILE;
            const bf16_t* Vn = kv + (tile_v & (NSLOT - 1)) * 2 * TILE + TILE;
            float ex[RB][8];                     // unrounded probabilities: summed into l one step behind, packed pairwise two steps behind
            float ar[RB];                        // the exponent of the NEXT step (its FMA runs one step ahead of its transcendental)
            if constexpr (WITH_S && !PRE) { fr[0] = kfrag(Ks, h, 0); fr[1] = kfrag(Ks, h, 1); fr[2] = kfrag(Ks, h, 2); }
#pragma unroll
            for (int rt = 0; rt < RA; ++rt) ar[rt] = fmaf(st[bc][rt][0][0], scale, neg_m[rt]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if constexpr (W1_DBG != 5) {
                if (g + 3 < 8) { if constexpr (WITH_S) fr[(g + 3) & 3] = kfrag(Ks, h, g + 3); }
                else if constexpr (NEXT) fr[(g + 3) & 3] = vfrag(Vn, hv, g + 3 - 8);
                }
#pragma unroll
                for (int rt = 0; rt < RA; ++rt) {
                    if constexpr (WITH_S && W1_DBG != 3)
                        st[bn][rt][g & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g & 3], qf[rt][g >> 1], g < 2 ? f32x4_t{0, 0, 0, 0} : st[bn][rt][g & 1], 0, 0, 0);
                    if constexpr (W1_DBG == 1) ex[rt][g] = ar[rt] * 0.001f; else
                    ex[rt][g] = __builtin_amdgcn_exp2f(ar[rt]);
                    if (g + 1 < 8) ar[rt] = fmaf(st[bc][rt][(g + 1) >> 2][(g + 1) & 3], scale, neg_m[rt]);
                    if (g >= 1 && W1_DBG != 2) l_run[rt] += ex[rt][g - 1];          // (one step behind its exponential: no wait for the transcendental pipe)
                    if (g >= 2 && (g & 1) == 0) {
                        const bf16x2_t v2 = __builtin_convertvector(f32x2_t{ex[rt][g - 2], ex[rt][g - 1]}, bf16x2_t);
                        pk[rt][(g >> 1) - 1] = __builtin_bit_cast(unsigned, v2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int rt = 0; rt < RA; ++rt) {
                l_run[rt] += ex[rt][7];
                const bf16x2_t v2 = __builtin_convertvector(f32x2_t{ex[rt][6], ex[rt][7]}, bf16x2_t);
                pk[rt][3] = __builtin_bit_cast(unsigned, v2);
                // (anchors: P and l are only READ in later blocks, and LLVM's sink pass would move the whole exp chain down there, out of the MFMAs' shadow)
                asm volatile("" : "+v"(pk[rt][0]), "+v"(pk[rt][1]), "+v"(pk[rt][2]), "+v"(pk[rt][3]), "+v"(l_run[rt]));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // a unit that crosses the causal diagonal or the end of the keys: its invisible scores become -inf BEFORE phase B looks for the maximum.  Its own small block (a
        // wave-uniform branch around it): the phases exist once -- with a masked and an unmasked copy of phase B hipcc kept two copies of O alive (64 registers) and spilled
        auto mask_unit = [&](auto BX_, int tile_x, int hx) {
            constexpr int bx_ = decltype(BX_)::value;
#pragma unroll
            for (int rt = 0; rt < RA; ++rt) {
                const int lim = lim0[rt] - tile_x * KT;
                const int rel = (lim < 0 ? 0 : (lim > KT ? KT : lim)) - lq * 8 - hx * 32;          // element (t, r) = key hx*32 + lq*8 + t*4 + r of the tile is visible iff t*4 + r < rel
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!(t * 4 + r < rel)) st[bx_][rt][t][r] = -INFINITY;
            }
        };
        // B: O += V(tile, half h)^T P  beside  the running maximum / rescale factor of the unit in buffer BX.  WITH_X = false: no successor unit.  PRE / NEXT as in phase A
        //    (NEXT reads K(tile_k, hk)'s first three fragments).  Returns whether a row of this lane moved its maximum.
        auto phase_b = [&](auto BX_, auto WITH_X_, auto PRE_, auto NEXT_, int tile, int h, int tile_k, int hk) -> bool {
            constexpr int bx_ = decltype(BX_)::value;
            constexpr bool WITH_X = decltype(WITH_X_)::value, PRE = decltype(PRE_)::value, NEXT = decltype(NEXT_)::value;
            const bf16_t* Vt = kv + (tile & (NSLOT - 1)) * 2 * TILE + TILE;
            const bf16_t* Kn = kv + (tile_k & (NSLOT - 1)) * 2 * TILE;
            if constexpr (!PRE) { fr[0] = vfrag(Vt, h, 0); fr[1] = vfrag(Vt, h, 1); fr[2] = vfrag(Vt, h, 2); }
            bf16x8_t pf[RB];
#pragma unroll
            for (int rt = 0; rt < RA; ++rt) pf[rt] = __builtin_bit_cast(bf16x8_t, u32x4_t{pk[rt][0], pk[rt][1], pk[rt][2], pk[rt][3]});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if constexpr (W1_DBG != 5) {
                if (t + 3 < 8) fr[(t + 3) & 3] = vfrag(Vt, h, t + 3);
                else if constexpr (NEXT) fr[(t + 3) & 3] = kfrag(Kn, hk, t + 3 - 8);
                }
#pragma unroll
                for (int rt = 0; rt < RA; ++rt) {
                    if constexpr (W1_DBG != 4) oacc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[t & 3], pf[rt], oacc[rt][t], 0, 0, 0);
                    // the vector chain of this row tile, one link per step
                    if constexpr (WITH_X && W1_DBG != 7) {
                        const f32x4_t s0 = st[bx_][rt][0], s1 = st[bx_][rt][1];
                        if (t == 0) mxr[rt] = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
                        if (t == 1) mxr[rt] = fmaxf(fmaxf(mxr[rt], s0[3]), s1[0]);
                        if (t == 2) mxr[rt] = fmaxf(fmaxf(mxr[rt], s1[1]), s1[2]);
                        if (t == 3) mxr[rt] = fmaxf(mxr[rt], s1[3]);
                        if (t == 4) { const unsigned u = __float_as_uint(mxr[rt]); const auto a2 = __builtin_amdgcn_permlane16_swap(u, u, false, false); mxr[rt] = fmaxf(__uint_as_float(a2[0]), __uint_as_float(a2[1])); }
                        if (t == 5) { const unsigned u = __float_as_uint(mxr[rt]); const auto b2 = __builtin_amdgcn_permlane32_swap(u, u, false, false); mxr[rt] = fmaxf(__uint_as_float(b2[0]), __uint_as_float(b2[1])) * scale; }
                        if (t == 6) { mvd[rt] = mxr[rt] > m_run[rt] + ATTN_DEFER; alpha[rt] = mvd[rt] ? __builtin_amdgcn_exp2f(m_run[rt] - mxr[rt]) : 1.f; }          // (m_run = -inf -> 0)
                        if (t == 7) { m_run[rt] = mvd[rt] ? mxr[rt] : m_run[rt]; neg_m[rt] = m_run[rt] == -INFINITY ? 0.f : -m_run[rt]; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            bool any = false;
            if constexpr (WITH_X && W1_DBG != 7) {
#pragma unroll
                for (int rt = 0; rt < RA; ++rt) { asm volatile("" : "+v"(m_run[rt]), "+v"(neg_m[rt]), "+v"(alpha[rt])); any |= mvd[rt]; }          // (anchors, as in phase A)
            }
            return any;
        };
        auto rescale = [&]() {               // between B(u) and A(u + 1): every product of unit u is in, none of unit u + 1
#pragma unroll
            for (int rt = 0; rt < RA; ++rt) {
                l_run[rt] *= alpha[rt];
#pragma unroll
                for (int t = 0; t < 8; ++t) oacc[rt][t] *= alpha[rt];
            }
        };
        using TY = std::integral_constant<bool, true>; using TN = std::integral_constant<bool, false>;

        if (ntile > 0) {
            // (three stages went out above; tile 0 is complete once at most the two younger ones are outstanding)
            if constexpr (NPC == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (RA > 0) {
                // prologue: S(0), its maximum (the generic phases with the halves that have no predecessor switched off: zero P, zero sums)
#pragma unroll
                for (int rt = 0; rt < RA; ++rt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[rt][e] = 0u;
                    st[1][rt][0] = st[1][rt][1] = f32x4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
                phase_a(I0{}, TY{}, TN{}, TY{}, 0, 0, 0, 0);             // (exp2(-inf) = 0: the dummy predecessor contributes nothing)
                if (need_mask(0)) mask_unit(I0{}, 0, 0);
                const bool mv = phase_b(I0{}, TY{}, TY{}, TY{}, 0, 0, 0, 1);          // (P = 0: O stays 0)
                if (__builtin_amdgcn_ballot_w64(mv)) rescale();
            }
            for (int j = 0; j + 1 < ntile; ++j) {
                if constexpr (RA > 0) {
                    phase_a(I1{}, TY{}, TY{}, TY{}, j, 1, j, 0);                        // A(2j):  S(2j+1) beside P(2j)
                    if (need_mask(j)) mask_unit(I1{}, j, 1);
                    const bool mv = phase_b(I1{}, TY{}, TY{}, TN{}, j, 0, 0, 0);         // B(2j):  O += V P(2j) beside the maximum of unit 2j+1
                    if (__builtin_amdgcn_ballot_w64(mv)) rescale();
                }
                sync_tile(j + 1);
                if constexpr (RA > 0) {
                    phase_a(I0{}, TY{}, TN{}, TY{}, j + 1, 0, j, 1);                    // A(2j+1): S(2j+2) beside P(2j+1)  (behind the tile barrier: its first fragments are read here)
                    if (need_mask(j + 1)) mask_unit(I0{}, j + 1, 0);
                    const bool mv = phase_b(I0{}, TY{}, TY{}, TY{}, j, 1, j + 1, 1);     // B(2j+1)
                    if (__builtin_amdgcn_ballot_w64(mv)) rescale();
                }
            }
            if constexpr (RA > 0) {          // the last tile: its second unit has no successor
                const int j = ntile - 1;
                phase_a(I1{}, TY{}, TY{}, TY{}, j, 1, j, 0);
                if (need_mask(j)) mask_unit(I1{}, j, 1);
                const bool mv = phase_b(I1{}, TY{}, TY{}, TN{}, j, 0, 0, 0);
                if (__builtin_amdgcn_ballot_w64(mv)) rescale();
                phase_a(I0{}, TN{}, TN{}, TY{}, j, 1, j, 1);
                phase_b(I0{}, TN{}, TY{}, TN{}, j, 1, 0, 0);
            }
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (a split of fewer than three tiles still has its re-read stages in flight: no DMA may outlive the wave)
        // ---- epilogue (attn_gqa128_kernel's) ----
#pragma unroll
        for (int rt = 0; rt < RA; ++rt) {
            float l = l_run[rt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const int row = blk_first_row + (rt * WAVES + wave) * 16 + lr;
            if (row >= blk_end_row) continue;
            if (p.splits == 1) {
                const int tok = row / G, head = kvh * G + row % G;
                bf16_t* orow = (bf16_t*)p.out + (long long)tok * p.ldo + (long long)head * D;
                const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    s16x4_t o = {(short)f2bf(oacc[rt][t][0] * inv), (short)f2bf(oacc[rt][t][1] * inv), (short)f2bf(oacc[rt][t][2] * inv), (short)f2bf(oacc[rt][t][3] * inv)};
                    *reinterpret_cast<s16x4_t*>(orow + t * 16 + lq * 4) = o;
                }
            } else {
                const long long grow = (long long)kvh * rows_total + row;
                const long long nrows_all = (long long)gridDim.y * rows_total;
                float* wo = p.ws_o + ((long long)bz * nrows_all + grow) * D;
#pragma unroll
                for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4_t*>(wo + t * 16 + lq * 4) = oacc[rt][t];
                if (lq == 0) { float* wml = p.ws_ml + ((long long)bz * nrows_all + grow) * 2; wml[0] = m_run[rt]; wml[1] = l; }
            }
        }
    };
    static_assert(RT == 2, "row tiles per wave: the dispatch below covers 0 / 1 / 2");
    if (my_tiles >= 2) run(std::integral_constant<int, 2>{});
    else if (my_tiles == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
}
