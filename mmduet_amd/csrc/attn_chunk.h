// attn_chunk.h -- attn_gqa128_chunk_kernel: the LLM attention of a multi-frame chunk (>= 1024 stacked rows per kv head; head_dim 128, bf16, causal GQA with a query offset
// over the KV arena; Qwen2Attention.forward, transformers qwen2/modeling_qwen2.py:200-240).  Included by attn.hip (AttnP, xcd helpers, quad_lanes_max, ATTN_DEFER, ATS_*).
//
// The wave program is attn_gqa128_kernel<2, 4, 8>'s of round 3 (256 query rows per block = 8 waves x 2 row tiles, four-slot K / V ring fed by all eight waves, a tile =
// two barrier-separated segments -- A_i = P.V(i-1) + scores(i), B_i = softmax(i) -- with waves 4-7 one segment behind waves 0-3, so that on every SIMD one wave is in its MFMA
// segment while its partner is in its softmax segment).  What is new (round 5) is the DECOMPOSITION:
//
//   A chunk of 26 frames is 35 row blocks x 4 kv heads = 140 units of ~245 key tiles.  A grid of (unit, key split) blocks runs in rounds of 256: no split leaves 116 CUs idle,
//   and the 7 splits the cost model picked (980 blocks = 3.83 rounds) write and re-read SEVEN fp32 partials per row (2 x 130 MB per launch at 15 k keys).  Here the
//   (unit, key tile) space is linearised -- unit-major, a unit's tiles in key order -- and cut into 256 CONTIGUOUS ranges of equal length, one per CU (the stream-K idea): every
//   CU is busy for the whole launch, a unit is shared by at most ceil(T_unit / range) + 1 blocks (2-3 at 15 k keys: 2 x 52 MB of partials), and the first chunk of a stream
//   (units of 1 .. 21 tiles) is balanced instead of waiting for its longest unit.  A block walks its range unit by unit (a "segment" = one unit's tiles inside the range);
//   the units' tile counts ride in the kernel argument (a 129-entry prefix table built on the host); where the cuts fall is recomputed by every block and by the merge kernel
//   with the same integer formulas -- no markers, nothing to initialise: a unit that one block covers alone is written straight to the output, and the merge skips it.
//
// Rounding points = attn_gqa128_kernel's; a row's result depends on where its unit is cut (fp32 summation order of the partials), deterministically.
#pragma once

// unit geometry: start[bx] = key tiles of the units 0 .. bx - 1 of one kv head (start[qblocks] = tiles per kv head); built on the host once per launch, rides in the
// kernel argument (the attention kernel and the merge read it with scalar loads: recomputing it per wave cost the merge 280 us at 35 k rows)
#ifndef CHUNK_DBG
#define CHUNK_DBG 0          // timing-only ablations (wrong results): tools/probes/chunk_ablate.sh
#endif
constexpr int CHUNK_TAB = 128, CHUNK_UNITS = 512;
struct ChunkTab { int qblocks; int start[CHUNK_TAB + 1]; unsigned short b0[CHUNK_UNITS], b1[CHUNK_UNITS]; };          // b0 / b1[kvh * qblocks + bx]: first / last block that shares the unit
// block b of nb owns linear tiles [b * W / nb, (b + 1) * W / nb)
__device__ __forceinline__ long long chunk_cut(long long W, int nb, int b) { return (W * b) / nb; }
__global__ __launch_bounds__(512, 1) void attn_gqa128_chunk_kernel(AttnP p, ChunkTab tab) {
    constexpr int RT = 2, NSLOT = 4, WAVES = 8;
    constexpr int D = 128, KT = 64, TILE = KT * D;
    constexpr int NPC = 16 / WAVES;                              // (K piece, V^T piece) pairs per wave and tile
    extern __shared__ __attribute__((aligned(16))) bf16_t kv[];        // NSLOT slots of (K tile, V^T tile): 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int G = p.nh / p.nkv;
    const int rows_total = p.S * G;
    const int qblocks = tab.qblocks, W1 = tab.start[tab.qblocks];
    const long long W = (long long)W1 * p.nkv;
    const int nb = gridDim.x;
    // XCD-aware block order: consecutive linear ranges (the same unit's K / V) on the same XCD
    int blk = blockIdx.x;
    { const int xcd = blk & 7, q = nb >> 3, r = nb & 7; blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blk >> 3); }
    long long pos = chunk_cut(W, nb, blk);
    const long long pos_end = chunk_cut(W, nb, blk + 1);
    const long long n_ctx = p.n_ctx, n_tot = p.n_ctx + p.S;
    const long long nrows_all = (long long)p.nkv * rows_total;
    const int vsw = (lr >> 1) & 7;
    const int grp = wave >> 2;

    // per-lane parts of the DMA addresses (fixed for the kernel)
    unsigned koff[NPC], voff[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int pc = wave + WAVES * j;
        const int key = pc * 4 + (lane >> 4);
        koff[j] = (unsigned)((key * D + (((lane & 15) ^ ((((key >> 3) & 3) << 2) | (key & 3))) * 8)) * 2);
        const int dim = pc * 8 + (lane >> 3);
        voff[j] = (unsigned)(((dim << 6) + (((lane & 7) ^ ((dim >> 1) & 7)) * 8)) * 2);
    }

    // locate the unit that holds `pos`
    int kvh = (int)(pos / W1), bx = 0;
    { const int rel = (int)(pos - (long long)kvh * W1); while (bx + 1 < qblocks && tab.start[bx + 1] <= rel) ++bx; }
    long long ustart = (long long)kvh * W1 + tab.start[bx];
    int utiles = tab.start[bx + 1] - tab.start[bx];

    bool first_seg = true;
    while (pos < pos_end) {
        // ---- one segment: tiles [t0, t1) of unit (kvh, bx) ----
        const int t0 = (int)(pos - ustart);
        const int t1 = (int)min((long long)utiles, pos_end - ustart);
        const bool whole = t0 == 0 && t1 == utiles;                                   // this block covers the unit alone: straight to the output
        const int part = blk - (int)tab.b0[kvh * qblocks + bx];                        // partial slot = how many cuts lie inside the unit before this segment
        const bf16_t* Kg = (const bf16_t*)p.K + kvh * p.k_hs;
        const bf16_t* Vg = (const bf16_t*)p.V + kvh * p.v_hs;
        const int row_base = bx * 256 + wave * (16 * RT);
        const int blk_first_row = bx * 256;
        const int blk_last_row = min(blk_first_row + 255, rows_total - 1);
        const long long blk_limit = p.causal ? min(n_tot, n_ctx + (long long)(blk_last_row / G) + 1) : n_tot;
        const long long blk_min_limit = p.causal ? n_ctx + (long long)(blk_first_row / G) + 1 : n_tot;   // keys below this are visible to every row
        const long long kbeg = (long long)t0 * KT;
        const long long kend = min(blk_limit, (long long)t1 * KT);
        const int ntile = t1 - t0;
        if (!first_seg) __syncthreads();          // the previous segment's last P.V reads of the ring are done before this one's first tiles are staged
        first_seg = false;

        auto stage = [&](int slot, long long k0) {
            if (CHUNK_DBG == 9) return;
            bf16_t* ks = kv + slot * 2 * TILE;
            bf16_t* vt = ks + TILE;
            // (the tile's two bases as OPAQUE scalar pairs: left visible, hipcc re-associates base + tile offset + lane offset inside the tile loop and falls back to 64-bit
            //  per-lane addresses -- two v_lshl_add_u64 + two moves per DMA; opaque, every DMA is `global_load_lds_dwordx4 v_off32, s[base]`)
            unsigned long long kbu = (unsigned long long)(Kg + k0 * D), vbu = (unsigned long long)(Vg + (((k0 >> 6) * D) << 6));
            asm volatile("" : "+s"(kbu), "+s"(vbu));
            const char* kb = (const char*)kbu;
            const char* vb = (const char*)vbu;
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                const int pc = wave + WAVES * j;
                unsigned ko = koff[j], vo = voff[j];
                asm volatile("" : "+v"(ko), "+v"(vo));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko), (__attribute__((address_space(3))) void*)(ks + pc * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vo), (__attribute__((address_space(3))) void*)(vt + pc * 512), 16, 0, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < ntile) stage(t, kbeg + (long long)t * KT);          // (tile j >= 2 goes out during interval j - 2)

        int my_row[RT], my_tok[RT], my_head[RT]; bool row_ok[RT]; long long my_limit[RT];
        bf16x8_t qf[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            my_row[rt] = row_base + rt * 16 + lr;
            row_ok[rt] = my_row[rt] < rows_total;
            my_tok[rt] = row_ok[rt] ? my_row[rt] / G : 0;
            my_head[rt] = kvh * G + (row_ok[rt] ? my_row[rt] % G : 0);
            my_limit[rt] = !row_ok[rt] ? 0 : (p.causal ? n_ctx + my_tok[rt] + 1 : n_tot);
            const bf16_t* qrow = (const bf16_t*)p.q + (long long)my_tok[rt] * p.ldq + (long long)my_head[rt] * D;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (row_ok[rt]) v = *reinterpret_cast<const s16x8_t*>(qrow + c * 32 + lq * 8);
                qf[rt][c] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
        const bool wave_active = row_base < rows_total;                 // wave-uniform
        // (the q fragments are ordinary loads: hipcc's wait for them -- a vmcnt(0), which also drains the first two tiles' DMAs -- happens here, once per segment, not in the tile loop)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" :: "v"(qf[rt][c]));

        f32x4_t oacc[RT][8];
        float m_run[RT], l_run[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            m_run[rt] = -INFINITY; l_run[rt] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) oacc[rt][t] = f32x4_t{0, 0, 0, 0};
        }
        f32x4_t st[RT][2][2];
        bf16x8_t pf[RT][2];
        // start of global segment sg: this wave's pieces of tile k have landed (tile k + 1 may stay in flight), every LDS read of the previous interval is done, block barrier,
        // then (group 1 here, group 0 behind its softmax) tile k + 2 goes out into the slot of tile k - 2
        auto seg_barrier = [&](int k) {
            if (CHUNK_DBG != 8) {
            if (k < ntile) { if (k + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier(); }
            if (grp == 1 && k + 2 < ntile) stage((k + 2) & (NSLOT - 1), kbeg + (long long)(k + 2) * KT);
        };
        auto stage_after_softmax = [&](int i) { if (grp == 0 && i + 2 < ntile) stage((i + 2) & (NSLOT - 1), kbeg + (long long)(i + 2) * KT); };
        auto do_pv = [&](int i) {               // O += P(i) V(i)
            const bf16_t* Vt = kv + (i & (NSLOT - 1)) * 2 * TILE + TILE;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bf16x8_t vfb = *reinterpret_cast<const bf16x8_t*>(Vt + (t * 16 + lr) * KT + (((h * 4 + lq) ^ vsw) * 8));
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) oacc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfb, pf[rt][h], oacc[rt][t], 0, 0, 0);
                    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
        };
        auto do_qk = [&](int i) {               // S(i) = K(i) Q^T
            const bf16_t* Ks = kv + (i & (NSLOT - 1)) * 2 * TILE;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * D + (((c * 4 + lq) ^ lr) * 8));
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            st[rt][h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[rt][c], c == 0 ? f32x4_t{0, 0, 0, 0} : st[rt][h][t], 0, 0, 0);
                        if (c == 3) __builtin_amdgcn_sched_barrier(0);
                    }
        };
        // A_i, software-pipelined by hand: eight groups of 4 fragment reads + 8 MFMAs; group g + 1's reads are issued before group g's MFMAs, sched_barriers pin that order
        auto a_load = [&](auto G_, bf16x8_t (&fr)[4], const bf16_t* Vt, const bf16_t* Ks) {
            constexpr int g = decltype(G_)::value;
            if constexpr (g < 4) {
                constexpr int h = g >> 1, tq = (g & 1) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) fr[u] = CHUNK_DBG == 5 ? qf[0][u] : *reinterpret_cast<const bf16x8_t*>(Vt + ((tq + u) * 16 + lr) * KT + (((h * 4 + lq) ^ vsw) * 8));
            } else {
                constexpr int h = (g - 4) >> 1, c0 = ((g - 4) & 1) * 2;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    fr[u] = CHUNK_DBG == 6 ? qf[1][u] : *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + (lr >> 2) * 8 + (u & 1) * 4 + (lr & 3)) * D + ((((c0 + (u >> 1)) * 4 + lq) ^ lr) * 8));
            }
        };
        auto a_mma = [&](auto G_, const bf16x8_t (&fr)[4]) {
            constexpr int g = decltype(G_)::value;
            if constexpr (g < 4) {
                constexpr int h = g >> 1, tq = (g & 1) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) { if (CHUNK_DBG == 4) oacc[rt][tq + u][0] += (float)fr[u][0] * (float)pf[rt][h][0]; else oacc[rt][tq + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u], pf[rt][h], oacc[rt][tq + u], 0, 0, 0); }
            } else {
                constexpr int h = (g - 4) >> 1, c0 = ((g - 4) & 1) * 2;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int c = c0 + (u >> 1), t = u & 1;
                        if (CHUNK_DBG == 3) { if (c == 0) st[rt][h][t] = f32x4_t{0, 0, 0, 0}; st[rt][h][t][0] += (float)fr[u][0] * (float)qf[rt][c][0]; }
                        else st[rt][h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u], qf[rt][c], c == 0 ? f32x4_t{0, 0, 0, 0} : st[rt][h][t], 0, 0, 0);
                    }
            }
        };
        auto do_a = [&](int i) {                // P.V of tile i - 1, scores of tile i
            const bf16_t* Vt = kv + ((i - 1) & (NSLOT - 1)) * 2 * TILE + TILE;
            const bf16_t* Ks = kv + (i & (NSLOT - 1)) * 2 * TILE;
            bf16x8_t fa[4], fb[4];
#define CHUNK_A_STEP(g, cur, nxt) do { if constexpr ((g) + 1 < 8) a_load(std::integral_constant<int, ((g) + 1 < 8 ? (g) + 1 : 7)>{}, nxt, Vt, Ks); __builtin_amdgcn_sched_barrier(0); \
                                       a_mma(std::integral_constant<int, (g)>{}, cur); __builtin_amdgcn_sched_barrier(0); } while (0)
            a_load(std::integral_constant<int, 0>{}, fa, Vt, Ks);
            CHUNK_A_STEP(0, fa, fb); CHUNK_A_STEP(1, fb, fa); CHUNK_A_STEP(2, fa, fb); CHUNK_A_STEP(3, fb, fa); CHUNK_A_STEP(4, fa, fb); CHUNK_A_STEP(5, fb, fa); CHUNK_A_STEP(6, fa, fb); CHUNK_A_STEP(7, fb, fa);
#undef CHUNK_A_STEP
        };
        const int lim0[RT] = {(int)((my_limit[0] < kend ? my_limit[0] : kend) - kbeg), (int)((my_limit[RT - 1] < kend ? my_limit[RT - 1] : kend) - kbeg)};      // (a segment's key range fits 31 bits)
        auto do_softmax = [&](int i) {          // P(i) from S(i): ONE running-max decision per row and 64-key tile
            const int tq0 = i * KT;                                           // tile-relative to kbeg
            const bool need_mask = (kbeg + tq0 + KT > blk_min_limit) || (kbeg + tq0 + KT > kend);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int lim = lim0[rt] - tq0;
                const int rel = (lim < 0 ? 0 : (lim > KT ? KT : lim)) - lq * 8;
#define SV(h, t, r) st[rt][h][t][r]
                if (need_mask) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (!(h * 32 + t * 4 + r < rel)) SV(h, t, r) = -INFINITY;
                }
                float mx = CHUNK_DBG == 7 ? SV(0, 0, 0) : fmaxf(fmaxf(fmaxf(fmaxf(SV(0, 0, 0), SV(0, 0, 1)), fmaxf(SV(0, 0, 2), SV(0, 0, 3))), fmaxf(fmaxf(SV(0, 1, 0), SV(0, 1, 1)), fmaxf(SV(0, 1, 2), SV(0, 1, 3)))),
                                 fmaxf(fmaxf(fmaxf(SV(1, 0, 0), SV(1, 0, 1)), fmaxf(SV(1, 0, 2), SV(1, 0, 3))), fmaxf(fmaxf(SV(1, 1, 0), SV(1, 1, 1)), fmaxf(SV(1, 1, 2), SV(1, 1, 3)))));
                if (CHUNK_DBG != 7) mx = quad_lanes_max(mx);
                mx *= p.scale_log2;
                if (mx > m_run[rt] + ATTN_DEFER) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run[rt] - mx);
                    l_run[rt] *= alpha;
#pragma unroll
                    for (int t = 0; t < 8; ++t) oacc[rt][t] *= alpha;
                    m_run[rt] = mx;
                }
                const float neg_m = m_run[rt] == -INFINITY ? 0.f : -m_run[rt];
                float psum = 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    s16x8_t pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float a_ = fmaf(SV(h, e >> 2, e & 3), p.scale_log2, neg_m); float pv = CHUNK_DBG == 1 ? a_ : __builtin_amdgcn_exp2f(a_); if (CHUNK_DBG != 2) psum += pv; pk[e] = (short)f2bf(pv); }
                    pf[rt][h] = __builtin_bit_cast(bf16x8_t, pk);
                }
#undef SV
                l_run[rt] += psum;
            }
        };
        if (ntile > 0) {
            if (grp == 1) seg_barrier(0);
            if (grp == 0) seg_barrier(0);
            if (wave_active) do_qk(0);                          // A_0
            if (grp == 1) seg_barrier(1);
            if (wave_active) do_softmax(0);                     // B_0
            stage_after_softmax(0);
            for (int i = 1; i < ntile; ++i) {
                if (grp == 0) seg_barrier(i);
                if (wave_active) do_a(i);                       // A_i = P.V(i - 1) + scores(i)
                if (grp == 1) seg_barrier(i + 1);
                if (wave_active) do_softmax(i);                 // B_i
                stage_after_softmax(i);
            }
            if (grp == 0) seg_barrier(ntile);
            if (wave_active) do_pv(ntile - 1);                  // A_n
        }

        // ---- this segment's result: the unit's output rows (alone) or its partial in slot `part` ----
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float l = l_run[rt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            if (!row_ok[rt]) continue;
            if (whole) {
                bf16_t* orow = (bf16_t*)p.out + (long long)my_tok[rt] * p.ldo + (long long)my_head[rt] * D;
                const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    s16x4_t o = {(short)f2bf(oacc[rt][t][0] * inv), (short)f2bf(oacc[rt][t][1] * inv), (short)f2bf(oacc[rt][t][2] * inv), (short)f2bf(oacc[rt][t][3] * inv)};
                    *reinterpret_cast<s16x4_t*>(orow + t * 16 + lq * 4) = o;
                }
            } else {
                const long long grow = (long long)kvh * rows_total + my_row[rt];
                float* wo = p.ws_o + ((long long)part * nrows_all + grow) * D;
#pragma unroll
                for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4_t*>(wo + t * 16 + lq * 4) = oacc[rt][t];
                if (lq == 0) { float* wml = p.ws_ml + ((long long)part * nrows_all + grow) * 2; wml[0] = m_run[rt]; wml[1] = l; }
            }
        }
        // ---- next unit ----
        pos = ustart + t1;
        if (t1 == utiles) { ustart += utiles; ++bx; if (bx == qblocks) { bx = 0; ++kvh; } utiles = tab.start[bx + 1] - tab.start[bx]; }
    }
}

// merge of attn_gqa128_chunk_kernel's partials: one wave per row; the row's unit says how many blocks shared it (same integer formulas as the attention kernel).
__global__ __launch_bounds__(256) void attn_combine128_chunk_kernel(AttnP p, ChunkTab tab, int nrows_all, int nb) {
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grow >= nrows_all) return;
    const int G = p.nh / p.nkv, rows_total = p.S * G;
    const int row = grow % rows_total, kvh = grow / rows_total;
    const int bx = row >> 8;
    const int parts = (int)tab.b1[kvh * tab.qblocks + bx] - (int)tab.b0[kvh * tab.qblocks + bx] + 1;
    if (parts == 1) return;                      // its block wrote the output itself
    const int tok = row / G, head = kvh * G + row % G;
    float ms = -INFINITY, ls = 0.f;
    if (lane < parts) { const float* ml = p.ws_ml + ((long long)lane * nrows_all + grow) * 2; ms = ml[0]; ls = ml[1]; }
    const float M = wave_max(ms);
    const float Mu = M == -INFINITY ? 0.f : M;
    const float w = __builtin_amdgcn_exp2f(ms - Mu);              // 0 for lanes without a part
    const float L = wave_sum(w * ls);
    float a0 = 0.f, a1 = 0.f;
    const float* obase = p.ws_o + (long long)grow * 128 + lane * 2;
    {   // the first four parts (all of them at production contexts) as four INDEPENDENT loads: one round trip, fixed summation order
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = float2{0.f, 0.f}; if (u < parts) v[u] = *reinterpret_cast<const float2*>(obase + (long long)u * nrows_all * 128); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float wj = __shfl(w, u, 64); a0 += wj * v[u].x; a1 += wj * v[u].y; }
    }
    for (int j = 4; j < parts; ++j) {
        const float2 v = *reinterpret_cast<const float2*>(obase + (long long)j * nrows_all * 128);
        const float wj = __shfl(w, j, 64);
        a0 += wj * v.x; a1 += wj * v.y;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    bf16_t* orow = (bf16_t*)p.out + (long long)tok * p.ldo + (long long)head * 128;
    s16x2_t o = {(short)f2bf(a0 * inv), (short)f2bf(a1 * inv)};
    *reinterpret_cast<s16x2_t*>(orow + lane * 2) = o;
}

// host side of the same formulas: key tiles per unit, total, and the most blocks that share one unit
static int chunk_unit_tiles_host(int rows_total, int G, long long n_ctx, int S, int causal, int bx) {
    const int last_row = (bx * 256 + 255 < rows_total - 1) ? bx * 256 + 255 : rows_total - 1;
    const long long n_tot = n_ctx + S;
    long long lim = causal ? n_ctx + (long long)(last_row / G) + 1 : n_tot;
    if (lim > n_tot) lim = n_tot;
    return (int)((lim + 63) >> 6);
}
static int chunk_block_of_host(long long W, int nb, long long tile) {
    int b = (int)((tile * nb) / W);
    while (b + 1 < nb && (W * (b + 1)) / nb <= tile) ++b;
    while (b > 0 && (W * b) / nb > tile) --b;
    return b;
}
// linear tiles per block of the contiguous decomposition (the dispatcher's criterion: short ranges pay a q load / pipeline fill / partial per segment)
static long long chunk_tiles_total(const AttnArgs& a) {
    const int G = a.nh / a.nkv, rows_total = a.S * G, qblocks = cdiv(rows_total, 256);
    long long W1 = 0;
    for (int b = 0; b < qblocks; ++b) W1 += chunk_unit_tiles_host(rows_total, G, a.n_ctx, a.S, a.causal, b);
    return W1 * a.nkv;
}
static hipError_t launch_gqa128_chunk(AttnP& p, const AttnArgs& a, hipStream_t st) {
    const int G = a.nh / a.nkv, rows_total = a.S * G, qblocks = cdiv(rows_total, 256);
    if (qblocks > CHUNK_TAB || qblocks * a.nkv > CHUNK_UNITS) return hipErrorInvalidValue;
    ChunkTab tab; tab.qblocks = qblocks; tab.start[0] = 0;
    long long W1 = 0; int tmax = 0;
    for (int b = 0; b < qblocks; ++b) { const int t = chunk_unit_tiles_host(rows_total, G, a.n_ctx, a.S, a.causal, b); W1 += t; if (t > tmax) tmax = t; tab.start[b + 1] = (int)W1; }
    for (int b = qblocks + 1; b <= CHUNK_TAB; ++b) tab.start[b] = (int)W1;
    const long long W = W1 * a.nkv;
    int nb = 256; if (W < nb) nb = (int)W;
    const int nrows_all = a.nkv * rows_total;
    for (int k = 0; k < a.nkv; ++k)
        for (int b = 0; b < qblocks; ++b) {
            const long long us = (long long)k * W1 + tab.start[b];
            tab.b0[k * qblocks + b] = (unsigned short)chunk_block_of_host(W, nb, us);
            tab.b1[k * qblocks + b] = (unsigned short)chunk_block_of_host(W, nb, us + (tab.start[b + 1] - tab.start[b]) - 1);
        }
    // without a workspace every unit must be whole: one block per unit; with one, bound the parts per unit by the workspace
    int parts_max = (int)(tmax / (W / nb)) + 2;
    if (!a.ws || parts_max > 64 || (size_t)parts_max * nrows_all * (128 + 2) * sizeof(float) > a.ws_bytes) return hipErrorInvalidValue;          // (the caller falls back to attn_gqa128_kernel)
    p.splits = parts_max; p.kv_per_split = 0;
    p.ws_o = a.ws;
    p.ws_ml = a.ws + (size_t)parts_max * nrows_all * 128;
    static bool attr_set[64] = {};
    int dev = 0; hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_gqa128_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(attn_gqa128_chunk_kernel, dim3(nb), dim3(512), 4 * 32768, st, p, tab);
    hipLaunchKernelGGL(attn_combine128_chunk_kernel, dim3(cdiv(nrows_all, 4)), dim3(256), 0, st, p, tab, nrows_all, nb);
    return hipGetLastError();
}
