// attn.hip -- softmax attention for both halves of the path (gfx950).
//
//   * LLM: causal GQA attention with a query offset over the KV arena -- Qwen2Attention.forward +
//     create_causal_mask (transformers qwen2/modeling_qwen2.py:200-240, 372-388): query row s sits at position
//     n_ctx + s and sees keys 0..n_ctx+s; 28 q heads share 4 kv heads (x7).
//   * ViT: full (non-causal) MHSA of SiglipAttention.forward (siglip/modeling_siglip.py:273-307), head_dim 72.
//
// attn_mfma_kernel (bf16): flash-style, 4 waves x 16 query rows, 32-key tiles.  Query rows of the heads that share a
//   kv head are flattened into one row space (row = tok*G + g) so a K/V tile staged in LDS is used by the whole
//   group.  S^T = mfma(K, Q) puts a query row in lane&15, so the online-softmax state (m, l, rescale) is lane-local
//   plus two xor-shuffles; O^T = mfma(V^T, P) consumes P straight from the S^T accumulator registers
//   (k-slot order {g*4+j, 16+g*4+j} on both operands).  Long contexts are split over grid.z and merged by
//   attn_combine_kernel (flash-decoding).
// attn_simple_kernel (any dtype / head_dim): one wave per (row, head); the fp32 parity path and the cross-check of
//   the MFMA kernel.
// softmax statistics are fp32; P is rounded to the storage type before P.V (flash / sdpa semantics).
#include "common.h"
#include <cstdlib>

// defer-max threshold (log2 units) of the flash kernels: the running maximum moves only when it grows by more than this
#define ATTN_DEFER 4.0f

struct AttnP {
    const void* q; const void* K; const void* V; void* out; float* ws_o; float* ws_ml;
    long long ldq, ldo, k_hs, k_ts, v_hs, v_ts, q_bs, kv_bs, o_bs;
    long long n_ctx;
    int S, nh, nkv, d, causal, splits, kv_per_split;
    const StepState* dyn;   // graph-replayed decode: n_ctx / arena base / capacity are read from device memory
    int layer;
    int v_tr;          // V stored transposed in 64-token blocks: (tok, e) at ((tok>>6)*d + e)*64 + (tok&63) inside the head region
    float scale_log2;
    const float* slabs; int n_slabs; const void* qkv_bias; const float2* rope_tab;     // AttnArgs::qkv_slabs (attn_gqa128<1> only)
    int slab_rows;      // rows of one slab (the GEMV's M: >= S when the step carries other streams' rows too, mmd_round_multi)
    int block_rows;     // attn_gqa128_w1_kernel: query rows per block (multiple of 16)
    int nseg;           // > 0 (decode form only): grid.x = nseg independent streams of S rows each (consecutive rows of q / out / slabs / rope_tab), stream j's context, capacity
                        // and arena base in dyn[j] (AttnArgs::segs); partials [seg][split][rows]
};

static thread_local int g_last_form[2] = {0, 0};          // (diagnostic only: which form the most recent launch took, mmd_op_attention_last_form)

__device__ __forceinline__ long long v_off(const AttnP& p, long long tok, int e) {
    return p.v_tr ? (((tok >> 6) * p.d + e) << 6) + (tok & 63) : tok * p.v_ts + e;
}

// XCD-aware block order: hardware deals consecutive block ids round-robin over the 8 XCDs (private L2 each).  Remap so
// that an XCD owns a CONTIGUOUS run of logical ids: the query blocks that share one K/V range (fastest logical index)
// then run on the same XCD and hit its L2 instead of each fetching K/V from HBM (measured on the ViT attention:
// FETCH_SIZE 858 MB per launch, 5x the 161 MB qkv buffer, with the default order).  Bijective for any grid size.
__device__ __forceinline__ void xcd_block_id(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nblk = gx * gy * gridDim.z;
    int id = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const int xcd = id & 7, q = nblk >> 3, r = nblk & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    bx = id % gx; by = (id / gx) % gy; bz = id / (gx * gy);
}

// max over the four lanes {l, l^16, l^32, l^48} (the lq groups that share a query row in the S^T layout) without going through the LDS
// crossbar: v_permlane16_swap / v_permlane32_swap of a register with itself leave (rows 0,0,2,2 | 1,1,3,3) resp. (lower, lower | upper, upper),
// so one max each folds the halves.  Exact, same value in all four lanes -- what two ds_bpermute shuffles gave, at VALU latency.
__device__ __forceinline__ float quad_lanes_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m1 = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(m1);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void attn_simple_kernel(AttnP p) {
    int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    int G = p.nh / p.nkv, kvh = h / G;
    const T* q = (const T*)p.q + b * p.q_bs + (long long)s * p.ldq + (long long)h * p.d;
    const T* K = (const T*)p.K + b * p.kv_bs + kvh * p.k_hs;
    const T* V = (const T*)p.V + b * p.kv_bs + kvh * p.v_hs;
    long long n_keys = p.causal ? p.n_ctx + s + 1 : p.n_ctx + p.S;
    int d = p.d;
    float q0 = lane < d ? to_f<T>(q[lane]) : 0.f, q1 = lane + 64 < d ? to_f<T>(q[lane + 64]) : 0.f;
    float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;
    for (long long j = 0; j < n_keys; ++j) {
        const T* kr = K + j * p.k_ts;
        float part = (lane < d ? q0 * to_f<T>(kr[lane]) : 0.f) + (lane + 64 < d ? q1 * to_f<T>(kr[lane + 64]) : 0.f);
        float sc = wave_sum(part) * p.scale_log2;
        float mn = fmaxf(m, sc);
        float alpha = exp2f(m - mn), pj = exp2f(sc - mn);
        float pr = rnd<T>(pj);
        o0 = o0 * alpha + (lane < d ? pr * to_f<T>(V[v_off(p, j, lane)]) : 0.f);
        o1 = o1 * alpha + (lane + 64 < d ? pr * to_f<T>(V[v_off(p, j, lane + 64)]) : 0.f);
        l = l * alpha + pj;
        m = mn;
    }
    T* o = (T*)p.out + b * p.o_bs + (long long)s * p.ldo + (long long)h * p.d;
    if (lane < d) o[lane] = from_f<T>(o0 / l);
    if (lane + 64 < d) o[lane + 64] = from_f<T>(o1 / l);
}

// ------------------------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void attn_mfma_kernel(AttnP p) {
    constexpr int KT = 32;                    // keys per tile
    constexpr int KLD = DP + 8;               // K tile row stride (elements)
    constexpr int VLD = KT + 8;               // V^T row stride
    constexpr int NC = DP / 32;               // QK k-steps
    constexpr int DVT = DP / 16;              // max PV d-tiles
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[DP * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int G = p.nh / p.nkv;
    const int kvh = blockIdx.y % p.nkv, b = blockIdx.y / p.nkv;
    const int rows_total = p.S * G;
    const int row0 = blockIdx.x * 64;
    const int d = p.d;
    const int dvt = (d + 15) >> 4;

    const bf16_t* Kg = (const bf16_t*)p.K + b * p.kv_bs + kvh * p.k_hs;
    const bf16_t* Vg = (const bf16_t*)p.V + b * p.kv_bs + kvh * p.v_hs;

    // this lane's query row (as the B operand of S^T = K.Q^T the row index is lane&15)
    const int my_row = row0 + wave * 16 + lr;
    const bool row_ok = my_row < rows_total;
    const int my_tok = row_ok ? my_row / G : 0;
    const int my_head = kvh * G + (row_ok ? my_row % G : 0);
    const long long n_tot = p.n_ctx + p.S;
    const long long my_limit = !row_ok ? 0 : (p.causal ? p.n_ctx + my_tok + 1 : n_tot);   // keys [0, my_limit) visible

    bf16x8_t qf[NC];
    {
        const bf16_t* qrow = (const bf16_t*)p.q + b * p.q_bs + (long long)my_tok * p.ldq + (long long)my_head * d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            int e = c * 32 + lq * 8;
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row_ok && e + 8 <= d) v = *reinterpret_cast<const s16x8_t*>(qrow + e);
            qf[c] = __builtin_bit_cast(bf16x8_t, v);
        }
    }

    // key range of this block / split
    int last_row = min(row0 + 63, rows_total - 1);
    long long blk_limit = p.causal ? min(n_tot, p.n_ctx + (long long)(last_row / G) + 1) : n_tot;
    long long kbeg = (long long)blockIdx.z * p.kv_per_split;
    long long kend = min(blk_limit, kbeg + p.kv_per_split);

    f32x4_t oacc[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t) oacc[t] = f32x4_t{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;

    for (long long k0 = kbeg; k0 < kend; k0 += KT) {
        __syncthreads();
        // stage K [32][DP] and V^T [DP][32]; 16-byte global loads, zero fill outside (keys >= kend, dims >= d)
        for (int i = tid; i < KT * (DP / 8); i += 256) {
            int kr = i / (DP / 8), c = (i % (DP / 8)) * 8;
            long long key = k0 + kr;
            s16x8_t kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (key < kend && c + 8 <= d) {
                kv = *reinterpret_cast<const s16x8_t*>(Kg + key * p.k_ts + c);
                if (p.v_tr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = (short)Vg[v_off(p, key, c + e)];
                } else vv = *reinterpret_cast<const s16x8_t*>(Vg + key * p.v_ts + c);
            }
            *reinterpret_cast<s16x8_t*>(Ks + kr * KLD + c) = kv;
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(c + e) * VLD + kr] = (bf16_t)vv[e];
        }
        __syncthreads();

        // S^T tiles: st[t][r] = S[q = lr][key = k0 + t*16 + lq*4 + r]
        f32x4_t st[2] = {f32x4_t{0, 0, 0, 0}, f32x4_t{0, 0, 0, 0}};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (t * 16 + lr) * KLD + c * 32 + lq * 8);
                st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c], st[t], 0, 0, 0);
            }
        float sv[8];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                long long key = k0 + t * 16 + lq * 4 + r;
                float v = (key < my_limit && key < kend) ? st[t][r] * p.scale_log2 : -INFINITY;
                sv[t * 4 + r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float m_new = fmaxf(m_run, mx);
        float m_use = m_new == -INFINITY ? 0.f : m_new;      // fully masked so far: keep everything at zero
        float alpha = exp2f(m_run - m_use);                  // m_run = -inf -> 0
        float psum = 0.f;
        s16x8_t pk;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float pv = exp2f(sv[i] - m_use);
            psum += pv;
            pk[i] = (short)f2bf(pv);
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pk);
#pragma unroll
        for (int t = 0; t < DVT; ++t) {
            if (t < dvt) {
                oacc[t] *= alpha;
                const bf16_t* vrow = Vt + (t * 16 + lr) * VLD;
                s16x4_t lo = *reinterpret_cast<const s16x4_t*>(vrow + lq * 4);
                s16x4_t hi = *reinterpret_cast<const s16x4_t*>(vrow + 16 + lq * 4);
                s16x8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), pf, oacc[t], 0, 0, 0);
            }
        }
    }

    // reduce l over the 4 lanes that share a query row
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);

    if (!row_ok) return;
    if (p.splits == 1) {
        bf16_t* orow = (bf16_t*)p.out + b * p.o_bs + (long long)my_tok * p.ldo + (long long)my_head * d;
        float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
        for (int t = 0; t < DVT; ++t) {
            int e = t * 16 + lq * 4;
            if (t < dvt && e + 4 <= d) {
                s16x4_t o = {(short)f2bf(oacc[t][0] * inv), (short)f2bf(oacc[t][1] * inv), (short)f2bf(oacc[t][2] * inv), (short)f2bf(oacc[t][3] * inv)};
                *reinterpret_cast<s16x4_t*>(orow + e) = o;
            } else if (t < dvt) {
                for (int r = 0; r < 4; ++r) if (e + r < d) orow[e + r] = f2bf(oacc[t][r] * inv);
            }
        }
    } else {
        long long grow = ((long long)b * p.nkv + kvh) * rows_total + my_row;       // global row id
        float* wo = p.ws_o + ((long long)blockIdx.z * gridDim.y * rows_total + grow) * DP;
#pragma unroll
        for (int t = 0; t < DVT; ++t) if (t < dvt) *reinterpret_cast<f32x4_t*>(wo + t * 16 + lq * 4) = oacc[t];
        if (lq == 0) {
            float* wml = p.ws_ml + ((long long)blockIdx.z * gridDim.y * rows_total + grow) * 2;
            wml[0] = m_run; wml[1] = l_run;
        }
    }
}

// merge split-KV partials: out = sum_s 2^(m_s - M) o_s / sum_s 2^(m_s - M) l_s
template <int DP>
__global__ void attn_combine_kernel(AttnP p, int nrows_total_all) {
    int grow = blockIdx.x;                       // ((b*nkv + kvh) * rows_total + row)
    int G = p.nh / p.nkv;
    int rows_total = p.S * G;
    int row = grow % rows_total, bk = grow / rows_total, kvh = bk % p.nkv, b = bk / p.nkv;
    int tok = row / G, head = kvh * G + row % G;
    float M = -INFINITY;
    for (int s = 0; s < p.splits; ++s) M = fmaxf(M, p.ws_ml[((long long)s * nrows_total_all + grow) * 2]);
    float L = 0.f;
    for (int s = 0; s < p.splits; ++s) {
        const float* ml = p.ws_ml + ((long long)s * nrows_total_all + grow) * 2;
        if (ml[0] != -INFINITY) L += exp2f(ml[0] - M) * ml[1];
    }
    bf16_t* orow = (bf16_t*)p.out + b * p.o_bs + (long long)tok * p.ldo + (long long)head * p.d;
    for (int e = threadIdx.x; e < p.d; e += blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < p.splits; ++s) {
            const float* ml = p.ws_ml + ((long long)s * nrows_total_all + grow) * 2;
            if (ml[0] != -INFINITY) acc += exp2f(ml[0] - M) * p.ws_o[((long long)s * nrows_total_all + grow) * DP + e];
        }
        orow[e] = f2bf(L > 0.f ? acc / L : 0.f);
    }
}

#ifdef MMDUET_ATTN_GRP_LSB
#define ATTN_GRP(w) ((w) & 1)
#else
#define ATTN_GRP(w) ((w) >> 2)
#endif
#ifdef MMDUET_ATTN_TIMING
// debug build only (`make ATTN_TIMING=1`, tools/attn_timing.py; never shipped): per-wave issue-time stamps of the chunk kernel's loop segments, summed over all
// waves of all launches since the last reset: [0] wait + barrier + next tile's DMA issue, [1] score MFMAs (+ K fragment reads), [2] softmax, [3] P.V MFMAs (+ V reads), [4] tiles, [5] whole kernel
__device__ unsigned long long g_attn_t[8];
extern "C" int mmd_debug_attn_timing(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_attn_t), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_t), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define ATS_DECL unsigned long long ats_[6] = {0, 0, 0, 0, 0, 0}, ats_t_ = __builtin_readcyclecounter(), ats_k_ = ats_t_
#define ATS(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); ats_[i] += n_ - ats_t_; ats_t_ = n_; } while (0)
#define ATS_FLUSH do { ats_[5] = __builtin_readcyclecounter() - ats_k_; if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_attn_t[i_], ats_[i_]); } while (0)
#else
#define ATS_DECL
#define ATS(i)
#define ATS_FLUSH
#endif

// ------------------------------------------------------------------------------------------------------------------
// attn_gqa128_kernel: the LLM attention of the streaming loop (head_dim 128, bf16, K arena [kvh][cap][128], V arena
// transposed in 64-token blocks).  One block = 4 waves x RT row-tiles of 16 "rows" (row = tok*G + g: the G query heads
// of a kv head are stacked so that a K/V tile is loaded once for all of them -- at S = 49, G = 7 the arithmetic
// intensity is ~340 flop per KV byte, i.e. this kernel is MFMA-bound, not KV-bandwidth-bound).
//   * 64-key tiles: K [64][128] and V^T [128][64] are each ONE contiguous 16 KB block of the arena, copied with
//     16-byte loads through registers into padded LDS images (conflict-free fragment reads); the next tile's global
//     loads are issued before the current tile's MFMAs (register prefetch) -- one tile of latency hiding.
//   * S^T = mfma(K, Q): the query row sits in lane&15, so running max / sum / rescale are lane-local (+2 xor-shuffles
//     across the 4 lanes that share a row); P goes from the S^T accumulators straight into the B operand of
//     O^T = mfma(V^T, P) (k-slot order {g*4+j, 16+g*4+j} on both operands) -- no LDS round trip for P.
//   * masking work only on tiles that cross the causal diagonal / the end of the key range.
//   * long contexts: grid.z splits the keys (flash-decoding); attn_combine128_kernel merges the fp32 partials.
// ------------------------------------------------------------------------------------------------------------------
//   * NSLOT = 4 (decode: one query block per kv head, <= 256 blocks, ONE per CU): a four-slot ring in 128 KB of LDS, three tiles in flight beyond the one being
//     consumed, counted vmcnt per tile, nt policy on the K / V stream (every byte is read once by one CU).  A 4-tile split at 15 k keys then pays ONE load
//     latency instead of four; the q prologue (slab sum + RoPE) loads are issued before the DMAs and finished under them.
//   * WAVES = 8 (chunks of >= 1024 rows per kv head): 256 query rows per block, ONE block per CU, the four-slot ring fed by all eight waves (4 DMA pieces per
//     wave and tile, counted vmcnt, raw barrier).  The 128-row form is bound by its K / V stream, not by MFMA or VALU: each block re-reads the whole key range
//     of its kv head from L2 with one 32 KB tile in flight (tools/attn_timing.py: 43-59 % of every wave's cycles sit in the per-tile wait + barrier; 2.1 GB per
//     launch at 15 k keys = 5.7 TB/s).  256 rows halve that traffic per flop and three tiles in flight cover the L2 latency.
template <int RT, int NSLOT = 2, int WAVES = 4>
// (launch bound: the four-wave forms are compiled for TWO blocks per CU although the decode ring's 128 KB of LDS admit one -- at one wave per SIMD hipcc has 512 registers, moves the
//  MFMA accumulators to AGPRs and pays a v_accvgpr move for every score it reads back: 276 of them and a dead 68-byte scratch frame in the decode form; bounded to 256 there are none)
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) void attn_gqa128_kernel(AttnP p) {
    constexpr int D = 128, KT = 64, TILE = KT * D;              // one K tile = one V^T tile = 8192 elements = 16 KB, contiguous in the arena
    constexpr int PF = NSLOT - 1;                                // tiles in flight beyond the one being consumed
    constexpr int BR = WAVES * 16 * RT;                          // query rows per block
    constexpr int NPC = 16 / WAVES;                              // (K piece, V^T piece) pairs per wave and tile
    constexpr bool ALLRING = NSLOT == 4 && WAVES == 8;           // every wave stages and computes; NSLOT == 4 && WAVES == 4 is the decode form (loaders + one compute wave)
    extern __shared__ __attribute__((aligned(16))) bf16_t kv[];        // NSLOT slots of (K tile, V^T tile): 64 / 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (a scalar: LDS-DMA destinations and the role split need no per-tile readfirstlane)
    const int lr = lane & 15, lq = lane >> 4;
    int bx, by, bz; xcd_block_id(bx, by, bz);
    const int G = p.nh / p.nkv, kvh = by;
    const int rows_total = p.S * G;
    if constexpr (RT == 1 && NSLOT == 4 && WAVES == 4) {
        if (p.nseg > 0) {                          // several streams' decode rows in one launch (mmd_round_multi): block column bx = stream; its rows, slabs and partials by offset
            const int seg = bx; bx = 0;
            p.dyn += seg;
            p.q = (const bf16_t*)p.q + (long long)seg * p.S * p.ldq;
            p.out = (bf16_t*)p.out + (long long)seg * p.S * p.ldo;
            if (p.slabs) { p.slabs += (long long)seg * p.S * ((p.nh + 2 * p.nkv) * D); p.rope_tab += seg * p.S * (D / 2); }
            const long long part = (long long)gridDim.z * gridDim.y * rows_total;
            p.ws_o += seg * part * D; p.ws_ml += seg * part * 2;
        }
    }
    const int row_base = bx * BR + wave * (16 * RT);
    long long n_ctx = p.n_ctx, k_hs = p.k_hs, v_hs = p.v_hs;
    const bf16_t* Kb = (const bf16_t*)p.K; const bf16_t* Vb = (const bf16_t*)p.V;
    int kv_per_split = p.kv_per_split;
    if (p.dyn) {                                   // captured decode step: everything that changes between replays lives in *dyn
        n_ctx = p.dyn->n_ctx;
        const long long cap = p.dyn->cap, le = (long long)p.nkv * cap * D;
        Kb = (const bf16_t*)p.dyn->K + p.layer * le; Vb = (const bf16_t*)p.dyn->V + p.layer * le;
        k_hs = v_hs = cap * D;
        const long long tiles = (n_ctx + p.S + 63) >> 6;
        kv_per_split = (int)((tiles + gridDim.z - 1) / gridDim.z) * 64;
    }
    const long long n_tot = n_ctx + p.S;
    const bf16_t* Kg = Kb + kvh * k_hs;
    const bf16_t* Vg = Vb + kvh * v_hs;

    int my_row[RT], my_tok[RT], my_head[RT]; bool row_ok[RT]; long long my_limit[RT];
    bf16x8_t qf[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        my_row[rt] = row_base + rt * 16 + lr;
        row_ok[rt] = my_row[rt] < rows_total;
        my_tok[rt] = row_ok[rt] ? my_row[rt] / G : 0;
        my_head[rt] = kvh * G + (row_ok[rt] ? my_row[rt] % G : 0);
        my_limit[rt] = !row_ok[rt] ? 0 : (p.causal ? n_ctx + my_tok[rt] + 1 : n_tot);
    }
    const bool wave_active = row_base < rows_total;                 // wave-uniform
    // key range of this block / split (multiples of 64 except at the very end)
    const int blk_first_row = bx * BR;
    const int blk_last_row = min(blk_first_row + BR - 1, rows_total - 1);
    const long long blk_limit = p.causal ? min(n_tot, n_ctx + (long long)(blk_last_row / G) + 1) : n_tot;
    const long long blk_min_limit = p.causal ? n_ctx + (long long)(blk_first_row / G) + 1 : n_tot;   // keys below this are visible to every row
    const long long kbeg = (long long)bz * kv_per_split;
    const long long kend = min(blk_limit, kbeg + kv_per_split);

    f32x4_t oacc[RT][8];
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        m_run[rt] = -INFINITY; l_run[rt] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) oacc[rt][t] = f32x4_t{0, 0, 0, 0};
    }

    // staging: both tiles are contiguous 16 KB runs of the arena and go to LDS by DMA (global_load_lds, 1 KiB pieces, 4 K + 4 V^T
    // per wave), double-buffered: tile i+1 is in flight while tile i is consumed, ONE barrier per tile, no staging registers.
    // DMA writes LDS lane-linearly, so the bank spreading is done on the SOURCE side: the 16-byte chunk q of K row `key` sits at
    // position q ^ f(key), f = ((key >> 3) & 3) * 4 + (key & 3), chunk q of V^T row `dim` at q ^ ((dim >> 1) & 7) -- conflict-free for the
    // b128 fragment reads.  MFMA row i of S^T tile t is key (i >> 2) * 8 + t * 4 + (i & 3) of the 32-key half (not t * 16 + i): a lane's
    // eight P values then belong to eight CONSECUTIVE keys, so its V^T operand is one 16-byte read instead of two 8-byte reads + repack.
    // The per-lane part of a DMA address (row / swizzle of this wave's four K and four V^T pieces) is fixed for the whole kernel: eight 32-bit byte offsets;
    // per tile only a scalar base moves (was: eight 64-bit address computations + eight readfirstlanes per tile, ~40 VALU of a VALU-bound loop).
    unsigned koff[NPC], voff[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int pc = wave + WAVES * j;
        const int key = pc * 4 + (lane >> 4);
        koff[j] = (unsigned)((key * (int)p.k_ts + (((lane & 15) ^ ((((key >> 3) & 3) << 2) | (key & 3))) * 8)) * 2);
        const int dim = pc * 8 + (lane >> 3);
        voff[j] = (unsigned)(((dim << 6) + (((lane & 7) ^ ((dim >> 1) & 7)) * 8)) * 2);
    }
    auto stage = [&](int slot, long long k0) {
        bf16_t* ks = kv + slot * 2 * TILE;
        bf16_t* vt = ks + TILE;
        const char* kb = (const char*)(Kg + k0 * p.k_ts);
        const char* vb = (const char*)(Vg + (((k0 >> 6) * D) << 6));
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int pc = wave + WAVES * j;
            unsigned ko = koff[j], vo = voff[j];
            asm volatile("" : "+v"(ko), "+v"(vo));          // (keeps the zero-extension next to the load: scalar base + 32-bit VGPR offset addressing, no 64-bit offset pairs carried through the loop)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko), (__attribute__((address_space(3))) void*)(ks + pc * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vo), (__attribute__((address_space(3))) void*)(vt + pc * 512), 16, 0, 0);
        }
    };
    const int vsw = (lr >> 1) & 7;
    const int ntile = kbeg < kend ? (int)((kend - kbeg + KT - 1) >> 6) : 0;

    if (RT == 1 && p.slabs) {
        // the new tokens' K rows / V columns: written by the block (query block 0 of this kv head) whose key range holds the position, BEFORE it stages
        // that tile.  Stores are write-through; vmcnt(0) + barrier (the loop's own, per tile) orders them before this block's DMA reads; nobody else reads them in this launch.
        if (bx == 0) {
            const int row_w = (p.nh + 2 * p.nkv) * D;
            const long long MN = (long long)p.slab_rows * row_w;
            bool in_first_tile = false;
            for (int tok = 0; tok < p.S; ++tok) {
                const long long pos = n_ctx + tok;
                if (pos < kbeg || pos >= kbeg + kv_per_split) continue;
                in_first_tile |= pos < kbeg + (long long)PF * KT;               // ... the tiles staged before the loop
                const bf16_t* bias = (const bf16_t*)p.qkv_bias;
                if (tid < 64) {                       // k: pair (i, i + 64)
                    const int i = tid, col = (p.nh + kvh) * D;
                    const float* sp = p.slabs + (long long)tok * row_w + col;
                    float q1[4], q2[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z) { q1[z] = z < p.n_slabs ? sp[z * MN + i] : 0.f; q2[z] = z < p.n_slabs ? sp[z * MN + i + 64] : 0.f; }
                    const float x1 = ((q1[0] + q1[1]) + q1[2]) + q1[3], x2 = ((q2[0] + q2[1]) + q2[2]) + q2[3];
                    const float a = bf2f(f2bf(x1 + bf2f(bias[col + i]))), b = bf2f(f2bf(x2 + bf2f(bias[col + i + 64])));
                    const float2 cs = p.rope_tab[tok * (D / 2) + i];
                    bf16_t* dst = const_cast<bf16_t*>(Kg) + pos * p.k_ts;
                    dst[i] = f2bf(bf2f(f2bf(a * cs.x)) + bf2f(f2bf(-b * cs.y)));
                    dst[i + 64] = f2bf(bf2f(f2bf(b * cs.x)) + bf2f(f2bf(a * cs.y)));
                } else if (tid < 192) {               // v: transposed 64-token blocks
                    const int i = tid - 64, col = (p.nh + p.nkv + kvh) * D;
                    const float* sp = p.slabs + (long long)tok * row_w + col;
                    float q1[4];
#pragma unroll
                    for (int z = 0; z < 4; ++z) q1[z] = z < p.n_slabs ? sp[z * MN + i] : 0.f;
                    const float x = ((q1[0] + q1[1]) + q1[2]) + q1[3];
                    const_cast<bf16_t*>(Vg)[(((pos >> 6) * D + i) << 6) + (pos & 63)] = f2bf(x + bf2f(bias[col + i]));
                }
            }
            // every later tile is staged behind the loop's own vmcnt(0) + barrier; only a position inside the FIRST tile must land before stage(0)
            if (in_first_tile) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }       // block-uniform
        }
    }
    // NSLOT == 4 (decode, <= 16 rows: wave 0 is the only one with rows): waves 1-3 are LOADERS -- they stage every tile (32 pieces dealt 11 / 11 / 10) and own the
    // counted waits; wave 0 issues no DMA at all, so the compiler's vmcnt(0) in front of its q prologue (slab sum + RoPE) covers the q loads only and the
    // prologue runs under the ring's first three tiles.  (A plain VGPR load next to LDS-DMAs in ONE wave always gets vmcnt(0) from hipcc.)
    const int ldr = wave - 1;
    // (per-lane byte offsets of the loader's 11 pieces, fixed for the kernel; per tile two opaque scalar bases: every DMA in the scalar-base + 32-bit-offset form -- the loaders'
    //  address arithmetic sits on the kernel's critical path, in front of the first three tiles)
    unsigned off3[11];
    if constexpr (NSLOT == 4 && !ALLRING) {
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int pp = ldr + 3 * j;
            if (pp < 16) {
                const int key = pp * 4 + (lane >> 4);
                off3[j] = (unsigned)((key * (int)p.k_ts + (((lane & 15) ^ ((((key >> 3) & 3) << 2) | (key & 3))) * 8)) * 2);
            } else {
                const int pc = pp - 16, dim = pc * 8 + (lane >> 3);
                off3[j] = (unsigned)(((dim << 6) + (((lane & 7) ^ ((dim >> 1) & 7)) * 8)) * 2);
            }
        }
    }
    auto stage3 = [&](int slot, long long k0) {
        bf16_t* ks = kv + slot * 2 * TILE;
        bf16_t* vt = ks + TILE;
        unsigned long long kbu = (unsigned long long)(Kg + k0 * p.k_ts), vbu = (unsigned long long)(Vg + (((k0 >> 6) * D) << 6));
        asm volatile("" : "+s"(kbu), "+s"(vbu));
        const char* kb = (const char*)kbu; const char* vb = (const char*)vbu;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int pp = ldr + 3 * j;
            unsigned o = off3[j];
            asm volatile("" : "+v"(o));
            if (pp < 16) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + o), (__attribute__((address_space(3))) void*)(ks + pp * 512), 16, 0, 2);
            } else if (pp < 32) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + o), (__attribute__((address_space(3))) void*)(vt + (pp - 16) * 512), 16, 0, 2);
            }
        }
    };
    if constexpr (ALLRING) {
#pragma unroll
        for (int t = 0; t < 2; ++t) if (t < ntile) stage(t, kbeg + (long long)t * KT);          // (tile j >= 2 goes out during interval j - 2, see the phase loop)
    } else if constexpr (NSLOT == 4) {
        if (wave != 0) {               // the loaders' whole life: one raw barrier per tile, matched by wave 0's below
#pragma unroll
            for (int t = 0; t < PF; ++t) if (t < ntile) stage3(t, kbeg + (long long)t * KT);
            for (int ti = 0; ti < ntile; ++ti) {
                // tile ti has landed once at most the younger tiles' DMAs (11 per tile from this wave, the third loader 10) are outstanding
                const int younger = min(PF - 1, ntile - 1 - ti);
                if (younger >= 2) { if (ldr < 2) asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
                else if (younger == 1) { if (ldr < 2) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (ti + PF < ntile) stage3((ti + PF) & (NSLOT - 1), kbeg + (long long)(ti + PF) * KT);       // the slot tile ti - 1 left: wave 0 is past it
            }
            return;
        }
    } else if (kbeg < kend) stage(0, kbeg);
    // q fragments AFTER the first tile's DMAs are in flight (their latency covers the q loads / the slab reduction + RoPE)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if (RT == 1 && p.slabs) {
            // q row from the projection's K-slabs: rnd(sum + bias), rotate-half RoPE with the unfused kernels' rounding points (ops.hip
            // slab_rope_append_kernel).  A lane's chunks c and c + 2 hold the partner elements i and i + 64.
            const int row_w = (p.nh + 2 * p.nkv) * D;
            const long long MN = (long long)p.slab_rows * row_w;
            const float* sp = p.slabs + (long long)my_tok[rt] * row_w + my_head[rt] * D + lq * 8;
            const bf16_t* bp = (const bf16_t*)p.qkv_bias + my_head[rt] * D + lq * 8;
            const float2* tp = p.rope_tab + my_tok[rt] * (D / 2) + lq * 8;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                s16x8_t o1 = {0, 0, 0, 0, 0, 0, 0, 0}, o2 = o1;
                if (row_ok[rt]) {
                    float x1[8], x2[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { x1[e] = sp[c * 32 + e]; x2[e] = sp[c * 32 + 64 + e]; }
                    for (int z = 1; z < p.n_slabs; ++z)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { x1[e] += sp[z * MN + c * 32 + e]; x2[e] += sp[z * MN + c * 32 + 64 + e]; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = bf2f(f2bf(x1[e] + bf2f(bp[c * 32 + e]))), b = bf2f(f2bf(x2[e] + bf2f(bp[c * 32 + 64 + e])));
                        const float2 cs = tp[c * 32 + e];
                        o1[e] = (short)f2bf(bf2f(f2bf(a * cs.x)) + bf2f(f2bf(-b * cs.y)));
                        o2[e] = (short)f2bf(bf2f(f2bf(b * cs.x)) + bf2f(f2bf(a * cs.y)));
                    }
                }
                qf[rt][c] = __builtin_bit_cast(bf16x8_t, o1); qf[rt][c + 2] = __builtin_bit_cast(bf16x8_t, o2);
            }
        } else {
            const bf16_t* qrow = (const bf16_t*)p.q + (long long)my_tok[rt] * p.ldq + (long long)my_head[rt] * D;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (row_ok[rt]) v = *reinterpret_cast<const s16x8_t*>(qrow + c * 32 + lq * 8);
                qf[rt][c] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
    }
    // (ring form: wave 0 wrote the new token's K row above and never counts vmcnt again -- drain it here, before the first barrier lets a loader stage that row's tile)
    if constexpr (NSLOT == 4 && !ALLRING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATS_DECL;
    if constexpr (ALLRING) {
        // The q fragments come from ordinary global loads; their first use sits behind a branch (wave_active), so hipcc would place its wait for them -- a full vmcnt(0),
        // which also drains the ring's DMAs -- at the score MFMAs INSIDE the tile loop, every tile.  Touch them here: the wait happens once, before the loop.
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" :: "v"(qf[rt][c]));
        // Phase-split loop.  Left to itself every wave runs score MFMAs -> softmax -> P.V MFMAs in the same order behind the same per-tile barrier, so the two waves
        // of a SIMD want the matrix pipe together and the VALU together (tools/attn_timing.py: ~6400 cycles per wave and tile for 1024 cycles of MFMA).  Here a
        // wave's tile i is two SEGMENTS,
        //     A_i = P.V of tile i-1  +  scores of tile i   (64 MFMAs, 32 LDS fragment reads)        B_i = softmax of tile i   (VALU only),
        // and waves 4-7 (the second wave of every SIMD: a block's waves are dealt to the SIMDs cyclically) run half a tile out of step with waves 0-3.
#ifdef MMDUET_ATTN_GRP_LSB
        const int grp = wave & 1;               // (experiment: pairs the waves that do NOT share a SIMD)
#else
        const int grp = wave >> 2;
#endif
        f32x4_t st[RT][2][2];
        bf16x8_t pf[RT][2];
        static_assert(!ALLRING || NPC == 2, "counted waits below assume 4 DMAs per wave and tile");
        // start of global segment sg: (even sg) this wave's pieces of tile sg / 2 have landed -- tile sg / 2 + 1 may stay in flight --, every LDS read of the
        // previous segment is done, block barrier, then (even sg) tile sg / 2 + 2 goes out
        // ONE barrier per tile.  Every wave runs the same sequence A_0 B_0 A_1 B_1 ... A_n; group 0 (waves 0-3) has barrier i in front of A_i, group 1 (waves 4-7) in
        // front of B_(i-1): between barriers i and i + 1 group 0 runs [A_i, B_i] and group 1 [B_(i-1), A_i] -- on every SIMD one wave is in its MFMA segment while the
        // partner is in its softmax segment, both do the same work per interval (no waiting for the longer of A / B), and both read tile i's K and tile i - 1's V in
        // interval i.  Ring: tile j is staged during interval j - 2 (group 1 right behind barrier j - 2, where its VALU segment starts; group 0 behind its softmax)
        // into the slot of tile j - 4, whose V was last read in interval j - 3; every wave waits for its own pieces of tile i -- tile i + 1 may stay in flight --
        // before barrier i.  A DMA issued beside bare MFMAs costs the wave 60-185 cycles, in a VALU stretch 25-60.
        auto seg_barrier = [&](int k) {
            if (k < ntile) { if (k + 1 < ntile) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            ATS(3);                                  // (debug build: [3] = the DMA wait alone, [0] = barrier + DMA issue)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (grp == 1 && k + 2 < ntile) stage((k + 2) & (NSLOT - 1), kbeg + (long long)(k + 2) * KT);
            ATS(0);
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
                    if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);          // (4 fragment reads + 8 MFMAs per group: bounds the fragments in flight and with them the register count)
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
        // A_i, software-pipelined by hand: eight groups of 4 fragment reads + 8 MFMAs (P.V: half h, output tiles 4q..4q+3; scores: half h, two 32-dim steps x both
        // key sub-tiles); group g + 1's reads are issued before group g's MFMAs and sched_barriers pin that order -- left alone the compiler hoists all 32 reads
        // (registers), fenced per group it exposes every LDS latency (measured: 2150 cycles for 1024 cycles of MFMA)
        auto a_load = [&](auto G, bf16x8_t (&fr)[4], const bf16_t* Vt, const bf16_t* Ks) {
            constexpr int g = decltype(G)::value;
            if constexpr (g < 4) {
                constexpr int h = g >> 1, tq = (g & 1) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) fr[u] = *reinterpret_cast<const bf16x8_t*>(Vt + ((tq + u) * 16 + lr) * KT + (((h * 4 + lq) ^ vsw) * 8));
            } else {
                constexpr int h = (g - 4) >> 1, c0 = ((g - 4) & 1) * 2;          // scores: half h, 32-dim steps c0, c0 + 1, both key sub-tiles -- fr[(c - c0) * 2 + t]
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    fr[u] = *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + (lr >> 2) * 8 + (u & 1) * 4 + (lr & 3)) * D + ((((c0 + (u >> 1)) * 4 + lq) ^ lr) * 8));
            }
        };
        auto a_mma = [&](auto G, const bf16x8_t (&fr)[4]) {
            constexpr int g = decltype(G)::value;
            if constexpr (g < 4) {
                constexpr int h = g >> 1, tq = (g & 1) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) oacc[rt][tq + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u], pf[rt][h], oacc[rt][tq + u], 0, 0, 0);
            } else {
                constexpr int h = (g - 4) >> 1, c0 = ((g - 4) & 1) * 2;          // four independent accumulation chains (rt x t) interleaved: a dependent MFMA is four issues behind its producer
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int c = c0 + (u >> 1), t = u & 1;
                        st[rt][h][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[u], qf[rt][c], c == 0 ? f32x4_t{0, 0, 0, 0} : st[rt][h][t], 0, 0, 0);
                    }
            }
        };
        auto do_a = [&](int i) {                // P.V of tile i - 1, scores of tile i
            const bf16_t* Vt = kv + ((i - 1) & (NSLOT - 1)) * 2 * TILE + TILE;
            const bf16_t* Ks = kv + (i & (NSLOT - 1)) * 2 * TILE;
            bf16x8_t fa[4], fb[4];
#define A_STEP(g, cur, nxt) do { if constexpr ((g) + 1 < 8) a_load(std::integral_constant<int, ((g) + 1 < 8 ? (g) + 1 : 7)>{}, nxt, Vt, Ks); __builtin_amdgcn_sched_barrier(0); \
                                 a_mma(std::integral_constant<int, (g)>{}, cur); __builtin_amdgcn_sched_barrier(0); } while (0)
            a_load(std::integral_constant<int, 0>{}, fa, Vt, Ks);
            A_STEP(0, fa, fb); A_STEP(1, fb, fa); A_STEP(2, fa, fb); A_STEP(3, fb, fa); A_STEP(4, fa, fb); A_STEP(5, fb, fa); A_STEP(6, fa, fb); A_STEP(7, fb, fa);
#undef A_STEP
        };
        const int lim0[RT] = {(int)((my_limit[0] < kend ? my_limit[0] : kend) - kbeg), (int)((my_limit[RT - 1] < kend ? my_limit[RT - 1] : kend) - kbeg)};      // (a split's key range fits 31 bits)
        auto do_softmax = [&](int i) {          // P(i) from S(i).  ONE running-max decision per row and 64-key tile (both halves' P wait for their P.V in the next segment: a
                                                // rescale between them would miss the first half), otherwise the arithmetic of the two-slot loop below.  (Both row tiles'
                                                // maxima first, one rescale branch, then all 32 exponentials -- more parallel on paper -- measured +10 %: kept row tile by row tile)
            const int t0 = i * KT;                                           // tile-relative to kbeg
            const bool need_mask = (kbeg + t0 + KT > blk_min_limit) || (kbeg + t0 + KT > kend);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int lim = lim0[rt] - t0;
                const int rel = (lim < 0 ? 0 : (lim > KT ? KT : lim)) - lq * 8;
#define SV(h, t, r) st[rt][h][t][r]
                if (need_mask) {                  // (in place: the scores are dead after this segment, and a copy would cost a move per element on the unmasked path)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (!(h * 32 + t * 4 + r < rel)) SV(h, t, r) = -INFINITY;
                }
                float mx = fmaxf(fmaxf(fmaxf(fmaxf(SV(0, 0, 0), SV(0, 0, 1)), fmaxf(SV(0, 0, 2), SV(0, 0, 3))), fmaxf(fmaxf(SV(0, 1, 0), SV(0, 1, 1)), fmaxf(SV(0, 1, 2), SV(0, 1, 3)))),
                                 fmaxf(fmaxf(fmaxf(SV(1, 0, 0), SV(1, 0, 1)), fmaxf(SV(1, 0, 2), SV(1, 0, 3))), fmaxf(fmaxf(SV(1, 1, 0), SV(1, 1, 1)), fmaxf(SV(1, 1, 2), SV(1, 1, 3)))));
                mx = quad_lanes_max(mx);
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
                    for (int e = 0; e < 8; ++e) { float pv = __builtin_amdgcn_exp2f(fmaf(SV(h, e >> 2, e & 3), p.scale_log2, neg_m)); psum += pv; pk[e] = (short)f2bf(pv); }
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
            ATS(1);
            if (grp == 1) seg_barrier(1);
            if (wave_active) do_softmax(0);                     // B_0
            stage_after_softmax(0);
            ATS(2);
            for (int i = 1; i < ntile; ++i) {
                if (grp == 0) seg_barrier(i);
                if (wave_active) do_a(i);                       // A_i = P.V(i - 1) + scores(i)
                ATS(1);
                if (grp == 1) seg_barrier(i + 1);
                if (wave_active) do_softmax(i);                 // B_i
                stage_after_softmax(i);
                ATS(2);
#ifdef MMDUET_ATTN_TIMING
                ats_[4] += 1;
#endif
            }
            if (grp == 0) seg_barrier(ntile);
            if (wave_active) do_pv(ntile - 1);                  // A_n
            ATS(3);
        }
    } else {
    int slot = 0, ti = 0;
    for (long long k0 = kbeg; k0 < kend; k0 += KT, slot = (slot + 1) & (NSLOT - 1), ++ti) {
        if constexpr (NSLOT == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (k0 + KT < kend) stage(slot ^ 1, k0 + KT);
            ATS(0);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (wave 0 only: its reads of the previous tile are done before the loaders may refill that slot)
            __builtin_amdgcn_s_barrier();
        }
        const bf16_t* Ks = kv + slot * 2 * TILE;
        const bf16_t* Vt = Ks + TILE;
        if (!wave_active) continue;
        const bool need_mask = (k0 + KT > blk_min_limit) || (k0 + KT > kend);
        // element (h, t, r) of this tile (key h*32 + lq*8 + t*4 + r) is visible to row rt iff h*32 + t*4 + r < rel[rt]
        int rel[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long lim = (my_limit[rt] < kend ? my_limit[rt] : kend) - k0;
            rel[rt] = (int)(lim < 0 ? 0 : (lim > KT ? KT : lim)) - lq * 8;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4_t st[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { st[rt][0] = f32x4_t{0, 0, 0, 0}; st[rt][1] = f32x4_t{0, 0, 0, 0}; }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + (lr >> 2) * 8 + t * 4 + (lr & 3)) * D + (((c * 4 + lq) ^ lr) * 8));
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) st[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[rt][c], st[rt][t], 0, 0, 0);
                }
            ATS(1);
            bf16x8_t pf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                // softmax on raw scores: the 1/sqrt(d)*log2(e) scale is folded into one FMA per element, and the running
                // maximum is only moved (O and l rescaled) when it grows by more than 2^DEFER -- P then ranges up to 2^DEFER
                // instead of 1, which bf16 P / fp32 O and l absorb; saves ~30 VALU per 16 MFMAs in a VALU-bound loop
                float sv[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[t * 4 + r] = st[rt][t][r];
                if (need_mask) {                  // wave-uniform: only tiles on the causal diagonal / at the end of the split pay for it
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (!(h * 32 + t * 4 + r < rel[rt])) sv[t * 4 + r] = -INFINITY;     // key = h*32 + lq*8 + t*4 + r, 32-bit, tile-relative
                }
                float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
                mx = quad_lanes_max(mx);
                mx *= p.scale_log2;                                   // scale > 0: max commutes with it
                if (mx > m_run[rt] + ATTN_DEFER) {                    // (first tile: m_run = -inf -> always)
                    const float alpha = __builtin_amdgcn_exp2f(m_run[rt] - mx);      // m_run = -inf -> 0
                    l_run[rt] *= alpha;
#pragma unroll
                    for (int t = 0; t < 8; ++t) oacc[rt][t] *= alpha;
                    m_run[rt] = mx;
                }
                const float neg_m = m_run[rt] == -INFINITY ? 0.f : -m_run[rt];
                float psum = 0.f;
                s16x8_t pk;
#pragma unroll
                for (int i = 0; i < 8; ++i) { float pv = __builtin_amdgcn_exp2f(fmaf(sv[i], p.scale_log2, neg_m)); psum += pv; pk[i] = (short)f2bf(pv); }
                l_run[rt] += psum;
                pf[rt] = __builtin_bit_cast(bf16x8_t, pk);
            }
            ATS(2);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bf16x8_t vfb = *reinterpret_cast<const bf16x8_t*>(Vt + (t * 16 + lr) * KT + (((h * 4 + lq) ^ vsw) * 8));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) oacc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfb, pf[rt], oacc[rt][t], 0, 0, 0);
            }
            ATS(3);
        }
#ifdef MMDUET_ATTN_TIMING
        ats_[4] += 1;
#endif
    }
    }
    if constexpr (RT == 2) { ATS_FLUSH; }

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float l = l_run[rt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!row_ok[rt]) continue;
        if (p.splits == 1) {
            bf16_t* orow = (bf16_t*)p.out + (long long)my_tok[rt] * p.ldo + (long long)my_head[rt] * D;
            float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                s16x4_t o = {(short)f2bf(oacc[rt][t][0] * inv), (short)f2bf(oacc[rt][t][1] * inv), (short)f2bf(oacc[rt][t][2] * inv), (short)f2bf(oacc[rt][t][3] * inv)};
                *reinterpret_cast<s16x4_t*>(orow + t * 16 + lq * 4) = o;
            }
        } else {
            long long grow = (long long)kvh * rows_total + my_row[rt];
            long long nrows_all = (long long)gridDim.y * rows_total;
            float* wo = p.ws_o + ((long long)bz * nrows_all + grow) * D;
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<f32x4_t*>(wo + t * 16 + lq * 4) = oacc[rt][t];
            if (lq == 0) { float* wml = p.ws_ml + ((long long)bz * nrows_all + grow) * 2; wml[0] = m_run[rt]; wml[1] = l; }
        }
    }
}

// merge the split partials of attn_gqa128_kernel: one wave per row.  The per-split (m, l) pairs are read one split per
// lane (a single latency instead of a serial chain), the weights live in registers and are broadcast with shuffles;
// the 128-dim partial rows are then accumulated with independent, fully coalesced 512-byte loads.
__global__ __launch_bounds__(256) void attn_combine128_kernel(AttnP p, int nrows_all) {
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (grow >= nrows_all) return;
    const int G = p.nh / p.nkv, rows_total = p.S * G;
    const int row = grow % rows_total, kvh = grow / rows_total;
    const int tok = row / G, head = kvh * G + row % G;
    float a0 = 0.f, a1 = 0.f, L = 0.f;
    float M = -INFINITY;
    for (int s0 = 0; s0 < p.splits; s0 += 64) {          // running merge over groups of 64 splits (usually one group)
        const int s = s0 + lane;
        float ms = -INFINITY, ls = 0.f;
        if (s < p.splits) { const float* ml = p.ws_ml + ((long long)s * nrows_all + grow) * 2; ms = ml[0]; ls = ml[1]; }
        float Mn = fmaxf(M, wave_max(ms));
        float Mu = Mn == -INFINITY ? 0.f : Mn;
        float resc = __builtin_amdgcn_exp2f(M - Mu);           // M = -inf -> 0
        float w = __builtin_amdgcn_exp2f(ms - Mu);              // 0 for empty splits
        a0 *= resc; a1 *= resc;
        L = L * resc + wave_sum(w * ls);
        M = Mn;
        const int cnt = min(64, p.splits - s0);
        const float* obase = p.ws_o + ((long long)s0 * nrows_all + grow) * 128 + lane * 2;
        // 32 independent 512-byte row loads in flight per wave (the merge is pure load latency: the 30 partials of a per-frame step are ONE round trip, not two),
        // fixed summation order
        for (int j0 = 0; j0 < cnt; j0 += 32) {
            float2 v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                v[u] = float2{0.f, 0.f};
                if (j0 + u < cnt) v[u] = *reinterpret_cast<const float2*>(obase + (long long)(j0 + u) * nrows_all * 128);
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                float wj = __shfl(w, (j0 + u) & 63, 64);
                a0 += wj * v[u].x; a1 += wj * v[u].y;
            }
        }
    }
    float inv = L > 0.f ? 1.0f / L : 0.f;
    bf16_t* orow = (bf16_t*)p.out + (long long)tok * p.ldo + (long long)head * 128;
    s16x2_t o = {(short)f2bf(a0 * inv), (short)f2bf(a1 * inv)};
    *reinterpret_cast<s16x2_t*>(orow + lane * 2) = o;
}

// the same merge for FEW rows (decode: 28 rows x up to 64 splits): one BLOCK per row, its 4 waves take a quarter of the splits each (one batch of <= 16
// independent 512-byte loads per wave instead of a chain of four) and meet in LDS.  7.0 -> ~4 us per layer at 15 k keys.
__global__ __launch_bounds__(256) void attn_combine128_rows_kernel(AttnP p, int nrows_all) {
    __shared__ float sm[4][130];
    int grow = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (p.nseg > 0) {                              // batched decode rows: nrows_all rows per stream, partials [seg][split][rows]
        const int seg = grow / nrows_all; grow -= seg * nrows_all;
        p.out = (bf16_t*)p.out + (long long)seg * p.S * p.ldo;
        const long long part = (long long)p.splits * nrows_all;
        p.ws_o += seg * part * 128; p.ws_ml += seg * part * 2;
    }
    const int G = p.nh / p.nkv, rows_total = p.S * G;
    const int row = grow % rows_total, kvh = grow / rows_total;
    const int tok = row / G, head = kvh * G + row % G;
    const int per = (p.splits + 3) >> 2;                       // <= 16
    const int s0 = wave * per, cnt = max(0, min(per, p.splits - s0));
    float ms = -INFINITY, ls = 0.f;
    if (lane < cnt) { const float* ml = p.ws_ml + ((long long)(s0 + lane) * nrows_all + grow) * 2; ms = ml[0]; ls = ml[1]; }
    const float Mw = wave_max(ms);
    const float Mu = Mw == -INFINITY ? 0.f : Mw;
    const float w = __builtin_amdgcn_exp2f(ms - Mu);            // 0 for empty / absent splits
    const float Lw = wave_sum(w * ls);
    float a0 = 0.f, a1 = 0.f;
    const float* obase = p.ws_o + ((long long)s0 * nrows_all + grow) * 128 + lane * 2;
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { v[u] = float2{0.f, 0.f}; if (u < cnt) v[u] = *reinterpret_cast<const float2*>(obase + (long long)u * nrows_all * 128); }
#pragma unroll
    for (int u = 0; u < 16; ++u) { const float wj = __shfl(w, u, 64); a0 += wj * v[u].x; a1 += wj * v[u].y; }
    sm[wave][lane * 2] = a0; sm[wave][lane * 2 + 1] = a1;
    if (lane == 0) { sm[wave][128] = Mw; sm[wave][129] = Lw; }
    __syncthreads();
    if (wave != 0) return;
    const float M = fmaxf(fmaxf(sm[0][128], sm[1][128]), fmaxf(sm[2][128], sm[3][128]));
    const float Mf = M == -INFINITY ? 0.f : M;
    float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sc = __builtin_amdgcn_exp2f(sm[k][128] - Mf);      // -inf -> 0
        L += sc * sm[k][129]; o0 += sc * sm[k][lane * 2]; o1 += sc * sm[k][lane * 2 + 1];
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    bf16_t* orow = (bf16_t*)p.out + (long long)tok * p.ldo + (long long)head * 128;
    s16x2_t o = {(short)f2bf(o0 * inv), (short)f2bf(o1 * inv)};
    *reinterpret_cast<s16x2_t*>(orow + lane * 2) = o;
}

#include "attn_w1.h"
#include "attn_chunk.h"

// attn_gqa128_w1_kernel (attn_w1.h): `qblocks` row blocks of block_rows (multiple of 16, <= 256) per kv head, `splits` key splits (the caller's choice)
static hipError_t launch_gqa128_w1(AttnP& p, const AttnArgs& a, hipStream_t st, int qblocks, int block_rows, int splits_hint) {
    const int G = a.nh / a.nkv, rows_total = a.S * G;
    const long long n_tot = a.n_ctx + a.S;
    const int tiles = cdiv(n_tot, 64);
    int splits = splits_hint < 1 ? 1 : splits_hint;
    if (!a.ws) splits = 1;
    if (splits > tiles) splits = tiles;
    if (splits > 64) splits = 64;
    while (splits > 1 && (size_t)splits * a.nkv * rows_total * (128 + 2) * sizeof(float) > a.ws_bytes) --splits;
    const int per = cdiv(tiles, splits) * 64;
    splits = cdiv(n_tot, per);
    p.splits = splits; p.kv_per_split = per; p.block_rows = block_rows;
    const int nrows_all = a.nkv * rows_total;
    p.ws_o = a.ws;
    p.ws_ml = a.ws ? a.ws + (size_t)splits * nrows_all * 128 : nullptr;
    static bool attr_set[64] = {};
    int dev = 0; hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_gqa128_w1_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((attn_gqa128_w1_kernel<2, 8>), dim3(qblocks, a.nkv, splits), dim3(512), 4 * 32768, st, p);
    if (splits > 1) hipLaunchKernelGGL(attn_combine128_kernel, dim3(cdiv(nrows_all, 4)), dim3(256), 0, st, p, nrows_all);
    return hipGetLastError();
}
// geometry for `rows_total` stacked rows per kv head: the fewest row blocks (<= 256 rows each), balanced in whole 16-row tiles; key splits fill the chip
// (one block per CU) -- but never below 2 tiles per split.
static hipError_t launch_gqa128_w1_auto(AttnP& p, const AttnArgs& a, hipStream_t st) {
    const int G = a.nh / a.nkv, rows_total = a.S * G;
    const int row_tiles = cdiv(rows_total, 16);
    const int qblocks = cdiv(row_tiles, 16);
    const int block_rows = cdiv(row_tiles, qblocks) * 16;
    const int tiles = cdiv(a.n_ctx + a.S, 64);
    // key splits by the cost model of launch_gqa128 (unit: one key tile of one resident block): rounds of 256 blocks, ~2 tiles of prologue / epilogue per block,
    // and the merge reads splits x rows x 520 B
    const int blocks = qblocks * a.nkv;
    const double rows_all = (double)a.nkv * rows_total;
    int maxs = tiles / 2; if (maxs < 1) maxs = 1; if (maxs > 64) maxs = 64;
    int splits = 1; double best = 1e30;
    for (int sp = 1; sp <= maxs; ++sp) {
        const double rounds = (double)cdiv((long long)blocks * sp, 256);
        const double cost = rounds * ((double)cdiv(tiles, sp) + 2.0) + (sp > 1 ? sp * rows_all * 6.7e-5 + 1.5 : 0.0);
        if (cost < best - 1e-9) { best = cost; splits = sp; }
    }
    return launch_gqa128_w1(p, a, st, qblocks, block_rows, splits);
}

template <int RT>
static hipError_t launch_gqa128(AttnP& p, const AttnArgs& a, hipStream_t st) {
    const int G = a.nh / a.nkv, rows_total = a.S * G;
    // chunks (>= 1024 rows per kv head): 256-row blocks, one per CU, four-slot ring
    const bool chunk8 = RT == 2 && rows_total >= 1024;
    const int qblocks = cdiv(rows_total, chunk8 ? 256 : 64 * RT);
    const int resident = chunk8 ? 256 : 512;                       // blocks the chip holds at once
    const long long n_tot = a.n_ctx + a.S;
    const int blocks = qblocks * a.nkv;
    const int tiles = cdiv(n_tot, 64);
    int splits = 1;
    if (a.dyn) {
        splits = a.dyn_splits;                                     // fixed grid under graph replay; empty splits write (m = -inf, l = 0)
    } else if (rows_total <= 16 && a.ws) {
        splits = 64;                                               // decode: same key ranges as the graph-replayed form -> identical bits
        if (splits > tiles) splits = tiles;
    } else if (a.ws) {
        // split-KV choice by a small cost model (units: one key tile of one block with two blocks resident per CU, ~2.6 us): the grid
        // runs in rounds of 512 resident blocks, a block costs its tiles plus ~2 tiles of prologue, and the merge kernel reads
        // splits x rows x 520 B.  Measured against it at S = 49 / 392 / 1274, n = 15 k: picks 42 / 5 / 7 splits (33 / 152 / 440 us; the
        // old fixed target of 576 blocks gave 42 / 202 / 485 us).
        const double rows_all = (double)a.nkv * rows_total;
        int maxs = tiles / 2; if (maxs < 1) maxs = 1; if (maxs > 64) maxs = 64;
        while (maxs > 1 && (size_t)maxs * a.nkv * rows_total * (128 + 2) * sizeof(float) > a.ws_bytes) --maxs;
        double best = 1e30;
        for (int sp = 1; sp <= maxs; ++sp) {
            const double rounds = (double)cdiv((long long)blocks * sp, resident);          // (a 256-row block's tile costs about what two resident 128-row blocks' tiles do: swept 1.0 .. 2.6, flat)
            const double cost = rounds * ((double)cdiv(tiles, sp) + 2.0) + (sp > 1 ? sp * rows_all * 6.7e-5 + 1.5 : 0.0);
            if (cost < best - 1e-9) { best = cost; splits = sp; }
        }
    }
    if constexpr (RT == 2) {
        // chunks: the contiguous (stream-K) decomposition of attn_chunk.h where the (unit, key split) grid is a bad fit -- it needs splits to fill the chip (each split is one more
        // fp32 partial per row) or leaves a fifth of it idle -- and a block's range is long enough to carry its per-segment costs (>= 14 tiles; 40+ units).  Measured against
        // the grid form over S = 147 .. 1911, 0 .. 30 k keys (profiles/r05_attention_chunk.md): 1.04 - 1.39 x where this rule takes it, the grid form elsewhere.
        if (chunk8 && a.variant == 0 && !a.dyn && a.ws && blocks >= 40) {
            const double eff = (double)blocks * splits / ((double)cdiv((long long)blocks * splits, resident) * resident);
            if (chunk_tiles_total(a) / 256 >= 14 && !(splits == 1 && eff >= 0.8)) {
                const hipError_t e = launch_gqa128_chunk(p, a, st);
                if (e == hipSuccess) { g_last_form[0] = 8; return e; }
                (void)hipGetLastError();
            }
        }
    }
    int per = cdiv(tiles, splits) * 64;
    if (!a.dyn) splits = cdiv(n_tot, per);
    p.splits = splits; p.kv_per_split = per;
    const int nrows_all = a.nkv * rows_total;
    p.ws_o = a.ws;
    p.ws_ml = a.ws ? a.ws + (size_t)splits * nrows_all * 128 : nullptr;
    // decode geometry (one query block per kv head, at most one block per CU): the four-slot ring; MMDUET_ATTN_DECODE_RING=0 keeps the two-slot form (A/B switch)
    static const bool ring_off = getenv("MMDUET_ATTN_DECODE_RING") && atoi(getenv("MMDUET_ATTN_DECODE_RING")) == 0;
    if (RT == 1 && qblocks == 1 && a.nkv * splits <= 256 && rows_total <= 16 && !ring_off) {
        static bool attr_set[64] = {};
        int dev = 0; hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_gqa128_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
            if (e != hipSuccess) return e;
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL((attn_gqa128_kernel<1, 4>), dim3(qblocks, a.nkv, splits), dim3(256), 4 * 32768, st, p);
    } else if (chunk8) {
        static bool attr8_set[64] = {};
        int dev = 0; hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !attr8_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_gqa128_kernel<2, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
            if (e != hipSuccess) return e;
            attr8_set[dev] = true;
        }
        hipLaunchKernelGGL((attn_gqa128_kernel<2, 4, 8>), dim3(qblocks, a.nkv, splits), dim3(512), 4 * 32768, st, p);
    } else
        hipLaunchKernelGGL((attn_gqa128_kernel<RT>), dim3(qblocks, a.nkv, splits), dim3(256), 2 * 32768, st, p);
    if (splits > 1) {
        if (nrows_all <= 64 && splits <= 64) hipLaunchKernelGGL(attn_combine128_rows_kernel, dim3(nrows_all), dim3(256), 0, st, p, nrows_all);
        else hipLaunchKernelGGL(attn_combine128_kernel, dim3(cdiv(nrows_all, 4)), dim3(256), 0, st, p, nrows_all);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// attn_rowmajor_kernel: MHSA over token-major K/V rows (the fused SigLIP qkv buffer; head_dim 72 -> 96 for QK^T, 80 for
// P.V).  Same register-resident softmax pipeline as attn_gqa128_kernel; the V^T operand comes from a ROW-MAJOR V tile via
// the gfx950 LDS transpose read (ds_read_b64_tr_b16): V is staged as [d/16][key/4][4 keys][16 dims] blocks, a 16-lane
// group reads one 4x16 block and every lane receives 4 consecutive keys of ONE dim -- exactly the k-slot order the P
// operand already has -- so no transposed copy of V is ever made.  128 query rows per block (4 waves x 2 row tiles),
// 64-key tiles, next tile prefetched into registers during the MFMAs.
// ------------------------------------------------------------------------------------------------------------------
template <int NC, int DVT, bool F16 = false, bool EXACT = false>      // NC = ceil(d/32) QK k-steps, DVT = max 16-wide output tiles (EXACT: ceil(d/16) == DVT, no per-tile `t < dvt` branches); F16: q / k / v / P / output are IEEE half (the fp16 tower)
__global__ __launch_bounds__(256, 2) void attn_rowmajor_kernel(AttnP p) {
    constexpr int RT = 2, KT = 64, DQ = NC * 32, KLD = DQ + 8;
    constexpr int NCH = DQ / 8;                                 // 16-byte chunks per key row (incl. zero padding)
    constexpr int PRE = (KT * NCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[DVT * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int d = p.d, dvt = EXACT ? DVT : (d + 15) >> 4, nch = d >> 3;
    const int G = p.nh / p.nkv;
    int bx, by, bz; xcd_block_id(bx, by, bz);
    const int head = by, kvh = head / G, b = bz;
    const int n_tot = (int)(p.n_ctx + p.S);
    const bf16_t* Kg = (const bf16_t*)p.K + b * p.kv_bs + kvh * p.k_hs;
    const bf16_t* Vg = (const bf16_t*)p.V + b * p.kv_bs + kvh * p.v_hs;
    const int row_base = bx * (64 * RT) + wave * (16 * RT);

    int my_tok[RT]; bool row_ok[RT]; int my_limit[RT];
    bf16x8_t qf[RT][NC];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        my_tok[rt] = row_base + rt * 16 + lr;
        row_ok[rt] = my_tok[rt] < p.S;
        my_limit[rt] = !row_ok[rt] ? 0 : (p.causal ? (int)p.n_ctx + my_tok[rt] + 1 : n_tot);
        const bf16_t* qrow = (const bf16_t*)p.q + b * p.q_bs + (long long)(row_ok[rt] ? my_tok[rt] : 0) * p.ldq + (long long)head * d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = c * 32 + lq * 8;
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row_ok[rt] && e + 8 <= d) v = *reinterpret_cast<const s16x8_t*>(qrow + e);
            qf[rt][c] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    const bool wave_active = row_base < p.S;
    const int blk_last = min(bx * (64 * RT) + 64 * RT - 1, p.S - 1);
    const int kend = p.causal ? min(n_tot, (int)p.n_ctx + blk_last + 1) : n_tot;
    const int blk_min_limit = p.causal ? (int)p.n_ctx + bx * (64 * RT) + 1 : n_tot;

    // zero both images once: padding chunks / dims are never written again
    for (int i = tid; i < KT * KLD / 8; i += 256) *reinterpret_cast<s16x8_t*>(Ks + i * 8) = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < DVT * 16 * 64 / 8; i += 256) *reinterpret_cast<s16x8_t*>(Vs + i * 8) = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};

    f32x4_t oacc[RT][DVT];
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        m_run[rt] = -INFINITY; l_run[rt] = 0.f;
#pragma unroll
        for (int t = 0; t < DVT; ++t) oacc[rt][t] = f32x4_t{0, 0, 0, 0};
    }

    // this thread's pieces of a tile -- (key, 16-byte chunk) pairs, KT * nch of them dealt round-robin -- are fixed for the whole kernel: source column, LDS
    // offsets and validity are computed once.  Loads are UNCONDITIONAL (a lane without a piece re-reads the last one, a key past the end reads the last valid
    // row: its scores are masked and its P is 0, any finite V row will do), so no per-tile zero fill of the staging registers and no branch around a load.
    s16x8_t pk_[PRE], pv_[PRE];
    int pkey[PRE], pcol[PRE], lk[PRE], lv[PRE]; bool pok[PRE];
    {
        const int pieces = KT * nch;
#pragma unroll
        for (int j = 0; j < PRE; ++j) {
            const int i = tid + 256 * j;
            pok[j] = i < pieces;
            const int ii = pok[j] ? i : pieces - 1;
            const int key = ii / nch, c = ii - key * nch;
            pkey[j] = key; pcol[j] = c * 8;
            lk[j] = key * KLD + c * 8;
            lv[j] = (((c >> 1) * 16 + (key >> 2)) * 4 + (key & 3)) * 16 + (c & 1) * 8;          // V image: block (dtile = c>>1, quad = key>>2), row key&3, cols (c&1)*8..
        }
    }
    const int kts = (int)p.k_ts, vts = (int)p.v_ts;
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < PRE; ++j) {
            const int row = min(k0 + pkey[j], kend - 1);
            pk_[j] = *reinterpret_cast<const s16x8_t*>(Kg + (unsigned)(row * kts + pcol[j]));          // (32-bit element offsets: launch_attention checks n_tot * stride < 2^31)
            pv_[j] = *reinterpret_cast<const s16x8_t*>(Vg + (unsigned)(row * vts + pcol[j]));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < PRE; ++j)
            if (pok[j]) {
                *reinterpret_cast<s16x8_t*>(Ks + lk[j]) = pk_[j];
                *reinterpret_cast<s16x8_t*>(Vs + lv[j]) = pv_[j];
            }
    };

    if (0 < kend) prefetch(0);
    for (int k0 = 0; k0 < kend; k0 += KT) {
        __syncthreads();
        commit();
        __syncthreads();
        if (k0 + KT < kend) prefetch(k0 + KT);
        if (!wave_active) continue;
        const bool need_mask = (k0 + KT > blk_min_limit) || (k0 + KT > kend);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4_t st[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { st[rt][0] = f32x4_t{0, 0, 0, 0}; st[rt][1] = f32x4_t{0, 0, 0, 0}; }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + t * 16 + lr) * KLD + c * 32 + lq * 8);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) st[rt][t] = mfma16<F16>(kf, qf[rt][c], st[rt][t]);
                }
            bf16x8_t pf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                // softmax on raw scores: the 1/sqrt(d)*log2(e) scale is folded into one FMA per element, and the running
                // maximum is only moved (O and l rescaled) when it grows by more than 2^DEFER -- P then ranges up to 2^DEFER
                // instead of 1, which bf16 P / fp32 O and l absorb; saves ~30 VALU per 16 MFMAs in a VALU-bound loop
                float sv[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[t * 4 + r] = st[rt][t][r];
                if (need_mask) {                  // wave-uniform BRANCH: only the tile at the end of the keys (and, causal, on the diagonal) pays the compares / selects
                    const int lim = min(my_limit[rt], kend) - (k0 + h * 32 + lq * 4);          // element (t, r) is visible iff t*16 + r < lim
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (!(t * 16 + r < lim)) sv[t * 4 + r] = -INFINITY;
                }
                float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
                mx = quad_lanes_max(mx);
                mx *= p.scale_log2;                                   // scale > 0: max commutes with it
                if (mx > m_run[rt] + ATTN_DEFER) {                    // (first tile: m_run = -inf -> always)
                    const float alpha = __builtin_amdgcn_exp2f(m_run[rt] - mx);      // m_run = -inf -> 0
                    l_run[rt] *= alpha;
#pragma unroll
                    for (int t = 0; t < DVT; ++t) oacc[rt][t] *= alpha;
                    m_run[rt] = mx;
                }
                const float neg_m = m_run[rt] == -INFINITY ? 0.f : -m_run[rt];
                float psum = 0.f;
                s16x8_t pk;
#pragma unroll
                for (int i = 0; i < 8; ++i) { float pv = __builtin_amdgcn_exp2f(fmaf(sv[i], p.scale_log2, neg_m)); psum += pv; pk[i] = (short)f2raw<F16>(pv); }
                l_run[rt] += psum;
                pf[rt] = __builtin_bit_cast(bf16x8_t, pk);
            }
#pragma unroll
            for (int t = 0; t < DVT; ++t) {
                if (t < dvt) {
                    // transpose reads: 4x16 blocks (quad = h*8 + lq) and (quad = h*8 + 4 + lq) of d-tile t
                    const bf16_t* blk = Vs + ((t * 16 + h * 8 + lq) * 64) + lr * 4;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(blk));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(blk + 4 * 64));
                    s16x8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bf16x8_t vfb = __builtin_bit_cast(bf16x8_t, vf);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) oacc[rt][t] = mfma16<F16>(vfb, pf[rt], oacc[rt][t]);
                }
            }
        }
    }

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float l = l_run[rt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!row_ok[rt]) continue;
        bf16_t* orow = (bf16_t*)p.out + b * p.o_bs + (long long)my_tok[rt] * p.ldo + (long long)head * d;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int t = 0; t < DVT; ++t) {
            const int e = t * 16 + lq * 4;
            if (t < dvt && e + 4 <= d) {
                s16x4_t o = {(short)f2raw<F16>(oacc[rt][t][0] * inv), (short)f2raw<F16>(oacc[rt][t][1] * inv), (short)f2raw<F16>(oacc[rt][t][2] * inv), (short)f2raw<F16>(oacc[rt][t][3] * inv)};
                *reinterpret_cast<s16x4_t*>(orow + e) = o;
            } else if (t < dvt) {
                for (int r = 0; r < 4; ++r) if (e + r < d) orow[e + r] = f2raw<F16>(oacc[rt][t][r] * inv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// attn_d72_ring_kernel: the SigLIP-so400m shape of attn_rowmajor_kernel (head_dim 72, bidirectional) with the K / V tiles brought in by LDS-DMA into two
// slots instead of through staging registers.  attn_rowmajor_kernel holds one tile in LDS and the next in 24 staging registers, commits them between two block
// barriers per tile and gathers V in 16-byte pieces from 32 different rows per load; it runs three blocks per CU at 146 registers.  Here tile i + 1 lands in the
// other slot while tile i is consumed, ONE raw s_barrier per tile both publishes tile i and retires the slot of tile i - 1, nothing is staged in registers
// (<= 128 registers, 40 KB of LDS: four blocks per CU), and both DMAs read 144 contiguous bytes per key.  187 -> 140-149 us per layer at 35 frames (MI355X, in the model).
// Same products, same summation order outside the matrix instructions as attn_rowmajor_kernel<3, 5, *, true>; the tower's output is the same to the bit on the seeded
// frames of tools/probes/vit_ring_ab.py (MMDUET_VIT_ATTN_RING=0 keeps the register-staged kernel).
//   K image: [64 keys][80] like V's (row stride 160 B = ten 16-byte places; the tenth re-reads the ninth chunk).  ds_read_b128 is served in four groups of 16 lanes that are
//            NOT lanes 0-15, 16-31, ...: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH LDS table) -- a group mixes rows lr of two lq values.  With ten places
//            per key the 16 places of every group are distinct mod 16 (lq = 0 rows fall on even places, lq = 1 rows on odd ones): conflict-free.  The unpadded [64][72] image of
//            round 3 (nine places per key, argued for lanes 0-15 as a group) put rows 13 and 4 + lq = 1 on the same banks: SQ_LDS_BANK_CONFLICT = 28 % of the LDS-active cycles.
//            Dims 0..63 take two 32-deep MFMAs, dims 64..71 one 16-deep MFMA (see qt below).
//   V image: [64 keys][80] ROW-MAJOR (row stride 160 B: the tenth 16-byte place of a key re-reads its ninth chunk -- output columns 72..79 are never stored).
//            ds_read_b64_tr_b16 takes one address per lane and transposes by lane position, so the 4-key x 16-dim block a 16-lane group wants need not be
//            contiguous: lane (key lr >> 2, dims 4 * (lr & 3)) points into the row-major rows (stride 40 dwords = 8 mod 32: the group's 16 x 8 bytes fall on
//            distinct banks).  With the blocked [d-tile][quad][4][16] image of attn_rowmajor_kernel a 1 KB DMA block touches 32 cache lines and uses a quarter
//            of each; row-major it touches 7-8 (measured: 153 -> 145 us).
// Waves 0-1 bring K, waves 2-3 V: five 1 KB blocks each per tile.
// What bounds it (profiles/r05_vit_attention.md, timing-only builds -DD72_DBG=n): the exponentials -- quarter rate on the vector port, a third of the matrix time at head_dim 72,
// -21 % without them; every other component is 3-9 %, and the savings add up (vector-issue-bound).  The ten transposing V reads of a half tile MUST be issued as one group:
// interleaved with the P.V MFMAs (hipcc does that as soon as the half tile is behind a branch) the kernel returns different results from run to run.
// ------------------------------------------------------------------------------------------------------------------
#ifndef D72_DBG
#define D72_DBG 0
#endif
// The 16-deep MFMA that finishes a score tile (dims 64..71) reads the 32-deep MFMA's result as SrcC.  Issued DIRECTLY behind its producer (hipcc does that whenever its schedule of
// the half tile shifts: twice this round, with a branch and with a scalar FMA in the softmax) the first row tile of every wave came back wrong and different from run to run -- with
// one independent MFMA between the two it is right (the order of the source: rt 0, rt 1, tail rt 0, tail rt 1).  The barriers below pin exactly that order of the matrix instructions
// (mask: everything but MFMAs may still cross); tests/test_attn_isa.py checks the distance in the listing.
#define D72_MFMA_PIN 0x7F6
// Round 6: the hazard is removed structurally -- the tail's 16-deep MFMA no longer takes the 32-deep MFMA's result as SrcC at all: it accumulates onto ZERO in registers of its own
// and one v_add_f32 per score register folds it in where the softmax reads the scores (D72_TAIL_SEPARATE 1).  No MFMA of the kernel then reads another MFMA's result of a different
// depth, in any instruction order; the pin and the ISA test stay as the second line.  0 = the chained form of round 5 (A/B: `make EXTRA=-DD72_TAIL_SEPARATE=0`).
#ifndef D72_TAIL_SEPARATE
#define D72_TAIL_SEPARATE 1
#endif
template <bool F16>
__global__ __launch_bounds__(256, 4) void attn_d72_ring_kernel(AttnP p) {
    constexpr int WAVES = 4, RT = 2, NC = 2, DVT = 5, D = 72, KT = 64, NSLOT = 2;
    constexpr int KLD = 80;                                           // K image row stride (elements)
    constexpr int NBLK = 10;                                          // 1 KB DMA blocks per image: K and V both [64][80]
    constexpr int NPW = NBLK / 2;                                     // ... per wave and tile: waves 0-1 bring K, waves 2-3 V
    constexpr int IMG = NBLK * 512;                                   // elements per image
    extern __shared__ __attribute__((aligned(16))) bf16_t ring72[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int G = p.nh / p.nkv;
    int bx, by, bz; xcd_block_id(bx, by, bz);
    const int head = by, kvh = head / G, b = bz;
    const int kend = (int)(p.n_ctx + p.S);
    const bf16_t* Kg = (const bf16_t*)p.K + b * p.kv_bs + kvh * p.k_hs;
    const bf16_t* Vg = (const bf16_t*)p.V + b * p.kv_bs + kvh * p.v_hs;
    const int row_base = bx * (16 * RT * WAVES) + wave * (16 * RT);

    int my_tok[RT]; bool row_ok[RT];
    bf16x8_t qf[RT][NC];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        my_tok[rt] = row_base + rt * 16 + lr;
        row_ok[rt] = my_tok[rt] < p.S;
        const bf16_t* qrow = (const bf16_t*)p.q + b * p.q_bs + (long long)(row_ok[rt] ? my_tok[rt] : 0) * p.ldq + (long long)head * D;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e = c * 32 + lq * 8;
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (row_ok[rt] && e + 8 <= D) v = *reinterpret_cast<const s16x8_t*>(qrow + e);
            qf[rt][c] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    // dims 64..71 go through ONE 16-deep MFMA (dims 64..79: 8-byte fragments, lanes lq >= 2 hold zeros) instead of a 32-deep one that is three quarters padding:
    // four registers fewer (what keeps the kernel at 128 and four blocks on a CU), a quarter fewer K bytes out of LDS for that step, half its matrix time
    s16x4_t qt[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        qt[rt] = s16x4_t{0, 0, 0, 0};
        if (row_ok[rt] && lq < 2) {
            const bf16_t* qrow = (const bf16_t*)p.q + b * p.q_bs + (long long)my_tok[rt] * p.ldq + (long long)head * D;
            qt[rt] = *reinterpret_cast<const s16x4_t*>(qrow + 64 + lq * 4);
        }
    }
    const bool wave_active = row_base < p.S;

    // this lane's five pieces of a tile: (key, byte column) of the 16 bytes that belong at its place of the wave's five 1 KB blocks
    const bool isK = wave < WAVES / 2;
    const int wsub = isK ? wave : wave - WAVES / 2;
    const char* src = (const char*)(isK ? Kg : Vg);
    const int ts2 = (int)(isK ? p.k_ts : p.v_ts) * 2;
    unsigned poff[NPW];                                           // byte offset of the piece inside a tile that starts at key 0
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
        const int ci = (wsub * NPW + u) * 64 + lane;                 // 0 .. 639
        const int key = ci / 10, c = ci - key * 10;                      // ten 16-byte places per key in BOTH images, the tenth (padding) re-reads the ninth chunk
        poff[u] = (unsigned)(key * ts2 + (c < 9 ? c : 8) * 16);
    }
    const unsigned last_row = (unsigned)((kend - 1) * ts2 + (D - 8) * 2);          // a key past the end re-reads the last chunk of the last row (finite; its scores are masked, its P is 0); row stride >= 144 B > 128
    auto stage = [&](int slot, int k0) {
        bf16_t* img = ring72 + slot * (2 * IMG) + (isK ? 0 : IMG);
        const unsigned t0 = (unsigned)(k0 * ts2);
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            unsigned off = min(t0 + poff[u], last_row);
            asm volatile("" : "+v"(off));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off), (__attribute__((address_space(3))) void*)(img + (wsub * NPW + u) * 512), 16, 0, 0);
        }
    };

    f32x4_t oacc[RT][DVT];
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        m_run[rt] = -INFINITY; l_run[rt] = 0.f;
#pragma unroll
        for (int t = 0; t < DVT; ++t) oacc[rt][t] = f32x4_t{0, 0, 0, 0};
    }

    const int ntile = (kend + KT - 1) >> 6;
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) if (i < ntile) stage(i, i * KT);
    // the q fragments are ordinary loads: hipcc waits for them with vmcnt(0), which would drain the ring wherever their first use lands -- make that here, once
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" :: "v"(qf[rt][c]));
    asm volatile("" :: "v"(qt[0])); asm volatile("" :: "v"(qt[1]));

    for (int i = 0; i < ntile; ++i) {
        const int k0 = i * KT;
        // this wave's five blocks of tile i have landed (nothing younger is in flight yet); the barrier publishes everybody's and retires the slot of tile i - 1
        if (D72_DBG != 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier(); }
        if (i + NSLOT - 1 < ntile) stage((i + NSLOT - 1) % NSLOT, k0 + (NSLOT - 1) * KT);
        if (!wave_active) continue;
        const bf16_t* Ks = ring72 + (i % NSLOT) * (2 * IMG);
        const bf16_t* Vs = Ks + IMG;
        const bool need_mask = k0 + KT > kend;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && k0 + 32 >= kend) break;          // the second half of the last tile lies wholly behind the keys (729 = 11 tiles + 25 keys): its P is 0, its exponentials -- the limiter of this kernel -- are not computed
            f32x4_t st[RT][2];
#if D72_TAIL_SEPARATE
            f32x4_t tt[RT][2];          // dims 64..71 of the scores: their own accumulators (see D72_TAIL_SEPARATE)
#endif
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { st[rt][0] = f32x4_t{0, 0, 0, 0}; st[rt][1] = f32x4_t{0, 0, 0, 0}; }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    bf16x8_t kf = D72_DBG == 6 ? qf[0][c] : *reinterpret_cast<const bf16x8_t*>(Ks + (h * 32 + t * 16 + lr) * KLD + c * 32 + lq * 8);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (D72_DBG == 3) { st[rt][t][0] += (float)kf[0] * (float)qf[rt][c][0]; } else st[rt][t] = mfma16<F16>(kf, qf[rt][c], st[rt][t]);
                        if (c == NC - 1) __builtin_amdgcn_sched_barrier(D72_MFMA_PIN);
                    }
                }
                // (rows lr >= 8 read the tail from the DUPLICATE in the tenth place: rows lr and lr + 8 are 1280 B = 0 banks apart, the duplicate sits 4 banks further -- the b64 reads of a
                //  32-lane group then fall on distinct banks; lq >= 2 reads the other copy's bytes: finite, against zeros)
                const s16x4_t kt = D72_DBG == 6 ? qt[0] : *reinterpret_cast<const s16x4_t*>(Ks + (h * 32 + t * 16 + lr) * KLD + 64 + ((lr >> 3) & 1) * 8 + lq * 4);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#if D72_TAIL_SEPARATE
                    const f32x4_t z = {0, 0, 0, 0};
                    if (D72_DBG == 3) { tt[rt][t] = z; tt[rt][t][1] = (float)kt[0]; } else if constexpr (F16) tt[rt][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, kt), __builtin_bit_cast(f16x4_t, qt[rt]), z, 0, 0, 0);
                    else tt[rt][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, qt[rt], z, 0, 0, 0);
#else
                    if (D72_DBG == 3) { st[rt][t][1] += (float)kt[0]; } else if constexpr (F16) st[rt][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, kt), __builtin_bit_cast(f16x4_t, qt[rt]), st[rt][t], 0, 0, 0);
                    else st[rt][t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, qt[rt], st[rt][t], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(D72_MFMA_PIN);
                }
            }
            bf16x8_t pf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float sv[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
#if D72_TAIL_SEPARATE
                    for (int r = 0; r < 4; ++r) sv[t * 4 + r] = st[rt][t][r] + tt[rt][t][r];
#else
                    for (int r = 0; r < 4; ++r) sv[t * 4 + r] = st[rt][t][r];
#endif
                if (need_mask) {                  // wave-uniform: the tile at the end of the keys only
                    int lim = (row_ok[rt] ? kend : 0) - (k0 + h * 32 + lq * 4);
                    asm volatile("" : "+v"(lim));             // (keeps this a BRANCH and the compares inside it: hipcc otherwise evaluates 7 compares + 8 selects per row tile in EVERY tile)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (!(t * 16 + r < lim)) sv[t * 4 + r] = -INFINITY;
                }
                // (a chain, not a tree: every step but the last folds into a three-input v_max3_f32 -- 4 instructions for the 8 values instead of 7; max is exact in any order)
                float mx = D72_DBG == 7 ? sv[0] : fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(sv[0], sv[1]), sv[2]), sv[3]), sv[4]), sv[5]), sv[6]), sv[7]);
                if (D72_DBG != 7) mx = quad_lanes_max(mx);
                mx *= p.scale_log2;
                if (mx > m_run[rt] + ATTN_DEFER) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run[rt] - mx);
                    l_run[rt] *= alpha;
#pragma unroll
                    for (int t = 0; t < DVT; ++t) oacc[rt][t] *= alpha;
                    m_run[rt] = mx;
                }
                const float neg_m = m_run[rt] == -INFINITY ? 0.f : -m_run[rt];
                float psum = 0.f;
                s16x8_t pk;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {          // (two scalar FMAs: v_pk_fma_f32 is the same fused operation per element but measures SLOWER here -- 147.6 vs 145.0 us per layer)
                    const float a0 = fmaf(sv[e], p.scale_log2, neg_m), a1 = fmaf(sv[e + 1], p.scale_log2, neg_m);
                    const float p0 = D72_DBG == 1 ? a0 : __builtin_amdgcn_exp2f(a0), p1 = D72_DBG == 1 ? a1 : __builtin_amdgcn_exp2f(a1);
                    if (D72_DBG != 2) { psum += p0; psum += p1; }
                    pk[e] = (short)f2raw<F16>(p0); pk[e + 1] = (short)f2raw<F16>(p1);
                }
                l_run[rt] += psum;
                pf[rt] = __builtin_bit_cast(bf16x8_t, pk);
            }
            // the five d-tiles' V^T fragments of this half tile, read BEHIND the softmax (20 registers that are then not live across it).  ds_read_b64_tr_b16 through the
            // builtin carries no memory operand, so hipcc waits vmcnt(0) in front of the first one of a tile: the next tile's DMA gets the score MFMAs + softmax of one half
            // tile to land instead of the whole tile.  With four blocks per CU that costs nothing measurable -- and the alternative, issuing the reads from an inline-asm
            // block the compiler cannot see into, gave run-to-run DIFFERENT results as soon as the score MFMAs were reordered (profiles/r03_attention_experiments.md): kept visible
            s16x4_t vlo[DVT], vhi[DVT];
            __builtin_amdgcn_sched_barrier(0);          // (the ten transpose reads stay ONE group between the softmax and the P.V MFMAs: see above)
            {
                const char __attribute__((address_space(3)))* vb = (const char __attribute__((address_space(3)))*)Vs + ((h * 32 + lq * 4 + (lr >> 2)) * 160 + (lr & 3) * 8);
#pragma unroll
                for (int t = 0; t < DVT; ++t) {
                    if (D72_DBG == 5) { vlo[t] = qt[0]; vhi[t] = qt[1]; continue; }
                    vlo[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vb + t * 32));
                    vhi[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vb + t * 32 + 2560));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < DVT; ++t) {
                const s16x4_t lo = vlo[t], hi = vhi[t];
                s16x8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bf16x8_t vfb = __builtin_bit_cast(bf16x8_t, vf);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) { if (D72_DBG == 4) { oacc[rt][t][0] += (float)vfb[0] * (float)pf[rt][0]; } else oacc[rt][t] = mfma16<F16>(vfb, pf[rt], oacc[rt][t]); }
            }
        }
    }

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float l = l_run[rt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!row_ok[rt]) continue;
        bf16_t* orow = (bf16_t*)p.out + b * p.o_bs + (long long)my_tok[rt] * p.ldo + (long long)head * D;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int t = 0; t < DVT; ++t) {
            const int e = t * 16 + lq * 4;
            if (e + 4 <= D) {
                s16x4_t o = {(short)f2raw<F16>(oacc[rt][t][0] * inv), (short)f2raw<F16>(oacc[rt][t][1] * inv), (short)f2raw<F16>(oacc[rt][t][2] * inv), (short)f2raw<F16>(oacc[rt][t][3] * inv)};
                *reinterpret_cast<s16x4_t*>(orow + e) = o;
            }
        }
    }
}

template <int NC, int DVT, bool EXACT = false>
static hipError_t launch_rowmajor(AttnP& p, const AttnArgs& a, hipStream_t st, bool f16 = false) {
    p.splits = 1; p.kv_per_split = 0;
    if (f16) hipLaunchKernelGGL((attn_rowmajor_kernel<NC, DVT, true, EXACT>), dim3(cdiv(a.S, 128), a.nh, a.batch), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_rowmajor_kernel<NC, DVT, false, EXACT>), dim3(cdiv(a.S, 128), a.nh, a.batch), dim3(256), 0, st, p);
    return hipGetLastError();
}

template <int DP>
static hipError_t launch_mfma(AttnP& p, const AttnArgs& a, hipStream_t st) {
    int G = a.nh / a.nkv;
    int rows_total = a.S * G;
    int qtiles = cdiv(rows_total, 64);
    int by = a.nkv * a.batch;
    long long n_tot = a.n_ctx + a.S;
    int blocks = qtiles * by;
    int splits = 1;
    if (blocks < 256) {
        int want = cdiv(512, blocks);
        int maxs = (int)(n_tot / 256); if (maxs < 1) maxs = 1;
        splits = want < maxs ? want : maxs;
        if (splits > 64) splits = 64;
        if (a.ws == nullptr) splits = 1;
        while (splits > 1 && (size_t)splits * by * rows_total * (DP + 2) * sizeof(float) > a.ws_bytes) --splits;
    }
    int per = (int)round_up(cdiv(n_tot, splits), 32);
    splits = cdiv(n_tot, per);
    p.splits = splits; p.kv_per_split = per;
    int nrows_all = by * rows_total;
    p.ws_o = a.ws;
    p.ws_ml = a.ws ? a.ws + (size_t)splits * nrows_all * DP : nullptr;
    hipLaunchKernelGGL((attn_mfma_kernel<DP>), dim3(qtiles, by, splits), dim3(256), 0, st, p);
    if (splits > 1) hipLaunchKernelGGL((attn_combine_kernel<DP>), dim3(nrows_all), dim3(64), 0, st, p, nrows_all);
    return hipGetLastError();
}

// Decode rows of SEVERAL streams in one launch (mmd_round_multi's talking streams): grid.x = stream, each stream's keys cut into the same number of splits so that
// nseg x nkv x splits blocks fill the chip once (one stream alone: 64 splits of ~4 tiles at 15 k keys; four streams: 16 splits of ~15 tiles -- a quarter of the
// launches, partial rows and merge work per token).  a = the FIRST stream's rows (q / out / slabs / rope_tab; the others follow at S-row steps); a.segs (device)
// holds every stream's context length, capacity and arena base.  bf16, head_dim 128, S x G <= 16.
hipError_t launch_attention_decode_multi(const AttnArgs& a, hipStream_t st) {
    const int G = a.nh / a.nkv, rows_total = a.S * G;
    if (a.nseg < 1 || a.nseg > 64 || !a.segs || a.d != 128 || !a.v_transposed || a.k_ts != 128 || rows_total > 16 || !a.ws || a.nseg * a.nkv > 256) return hipErrorInvalidValue;
    if (a.qkv_slabs && (a.n_slabs < 1 || a.n_slabs > 4)) return hipErrorInvalidValue;
    AttnP p;
    p.q = a.q; p.K = nullptr; p.V = nullptr; p.out = a.out; p.ldq = a.ldq; p.ldo = a.ldo;
    p.k_hs = 0; p.k_ts = a.k_ts; p.v_hs = 0; p.v_ts = a.v_ts; p.q_bs = 0; p.kv_bs = 0; p.o_bs = 0;
    p.n_ctx = 0; p.S = a.S; p.nh = a.nh; p.nkv = a.nkv; p.d = a.d; p.causal = 1; p.v_tr = 1; p.dyn = a.segs; p.layer = a.layer;
    p.scale_log2 = (1.0f / sqrtf((float)a.d)) * 1.4426950408889634f;
    p.slabs = a.qkv_slabs; p.n_slabs = a.n_slabs; p.qkv_bias = a.qkv_bias; p.rope_tab = (const float2*)a.rope_tab; p.slab_rows = a.slab_rows > 0 ? a.slab_rows : a.S * a.nseg;
    p.block_rows = 0; p.nseg = a.nseg;
    int splits = 256 / (a.nseg * a.nkv);
    if (splits > 64) splits = 64;
    if (splits < 2) splits = 2;                    // (the merge kernel writes the output: at least two partials keep one code path; > 32 streams per round never occur)
    const int nrows_all = a.nkv * rows_total;
    while (splits > 2 && (size_t)a.nseg * splits * nrows_all * (128 + 2) * sizeof(float) > a.ws_bytes) --splits;
    if ((size_t)a.nseg * splits * nrows_all * (128 + 2) * sizeof(float) > a.ws_bytes || a.nseg * a.nkv * splits > 512) return hipErrorInvalidValue;
    p.splits = splits; p.kv_per_split = 0;         // (per stream, computed in the kernel from its own context length -- the dyn path)
    p.ws_o = a.ws;
    p.ws_ml = a.ws + (size_t)a.nseg * splits * nrows_all * 128;
    static bool attr_set[64] = {};
    int dev = 0; hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_gqa128_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((attn_gqa128_kernel<1, 4>), dim3(a.nseg, a.nkv, splits), dim3(256), 4 * 32768, st, p);
    hipLaunchKernelGGL(attn_combine128_rows_kernel, dim3(a.nseg * nrows_all), dim3(256), 0, st, p, nrows_all);
    g_last_form[0] = 9; g_last_form[1] = splits;
    return hipGetLastError();
}

void attn_last_form(int* out2) { out2[0] = g_last_form[0]; out2[1] = g_last_form[1]; }          // (of the calling thread's most recent launch; model.hip copies it into the context right behind every LLM attention launch)
static hipError_t launch_attention_(int dtype, const AttnArgs& a, hipStream_t st, AttnP& p);
hipError_t launch_attention(int dtype, const AttnArgs& a, hipStream_t st) {
    if (a.S <= 0) return hipSuccess;
    AttnP p; p.splits = 1;
    g_last_form[0] = 0;
    const hipError_t e = launch_attention_(dtype, a, st, p);
    g_last_form[1] = p.splits;
    return e;
}
static hipError_t launch_attention_(int dtype, const AttnArgs& a, hipStream_t st, AttnP& p) {
    p.q = a.q; p.K = a.K; p.V = a.V; p.out = a.out; p.ws_o = nullptr; p.ws_ml = nullptr;
    p.ldq = a.ldq; p.ldo = a.ldo;
    p.k_hs = a.k_hs; p.k_ts = a.k_ts; p.v_hs = a.v_hs; p.v_ts = a.v_ts;
    p.q_bs = a.q_bstride; p.kv_bs = a.kv_bstride; p.o_bs = a.o_bstride;
    p.n_ctx = a.n_ctx; p.S = a.S; p.nh = a.nh; p.nkv = a.nkv; p.d = a.d; p.causal = a.causal;
    p.splits = 1; p.kv_per_split = 0; p.v_tr = a.v_transposed; p.dyn = a.dyn; p.layer = a.layer;
    p.scale_log2 = (1.0f / sqrtf((float)a.d)) * 1.4426950408889634f;
    p.slabs = a.qkv_slabs; p.n_slabs = a.n_slabs; p.qkv_bias = a.qkv_bias; p.rope_tab = (const float2*)a.rope_tab; p.slab_rows = a.slab_rows > 0 ? a.slab_rows : a.S; p.nseg = 0;
    const bool f16 = dtype == MMD_F16;          // the fp16 vision tower: the row-major kernel only
    bool can_mfma = (dtype == MMD_BF16 || f16) && (a.d % 8) == 0 && a.d <= 128 && (a.ldq % 8) == 0 && (a.k_ts % 8) == 0 && (a.v_ts % 8) == 0 &&
                    (a.k_hs % 8) == 0 && (a.v_hs % 8) == 0 && (a.kv_bstride % 8) == 0 && (a.q_bstride % 8) == 0;
    int variant = a.variant;
    const bool can_gqa128 = can_mfma && a.d == 128 && a.v_transposed && a.batch == 1 && a.k_ts == 128;
    // token-major K/V rows (ViT fused qkv): the transpose-read kernel; it keeps a whole sequence per (head, batch) block
    const bool can_rowmajor = can_mfma && !a.v_transposed && a.n_ctx + a.S < (1 << 30) && (a.o_bstride % 4) == 0 && (a.ldo % 4) == 0 &&
                              (a.n_ctx + a.S) * (a.k_ts > a.v_ts ? a.k_ts : a.v_ts) < (1ll << 31);          // (its staging loads address a sequence with 32-bit element offsets)
    if (variant == 0) variant = can_gqa128 ? 3 : (can_rowmajor && a.S >= 64 ? 4 : (can_mfma ? 2 : 1));
    if (a.qkv_slabs && variant != 3) return hipErrorInvalidValue;
    if (f16 && variant != 4) return hipErrorInvalidValue;
    if (variant == 4) {
        if (!can_rowmajor) return hipErrorInvalidValue;
        g_last_form[0] = 6;
        if (a.d <= 32) return launch_rowmajor<1, 2>(p, a, st, f16);
        if (a.d <= 64) return launch_rowmajor<2, 4>(p, a, st, f16);
        if (a.d == 72 && !a.causal && (a.k_ts % 8) == 0 && (a.v_ts % 8) == 0 && (a.k_hs % 8) == 0 && (a.v_hs % 8) == 0 && (a.kv_bstride % 8) == 0 &&
            ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.V % 16) == 0 && ((uintptr_t)a.q % 16) == 0 && (a.ldq % 8) == 0 && (a.q_bstride % 8) == 0 && a.n_ctx + a.S > 0) {
            // SigLIP-so400m, bidirectional: K / V tiles by LDS-DMA into two slots, four blocks per CU (MMDUET_VIT_ATTN_RING=0: the register-staged kernel, A/B)
            static const bool ring_off = getenv("MMDUET_VIT_ATTN_RING") && atoi(getenv("MMDUET_VIT_ATTN_RING")) == 0;
            if (!ring_off) {
                p.splits = 1; p.kv_per_split = 0; g_last_form[0] = 7;
                const size_t lds = (size_t)2 * 2 * 10 * 512 * sizeof(bf16_t);
                const dim3 grid(cdiv(a.S, 128), a.nh, a.batch);
                if (f16) hipLaunchKernelGGL((attn_d72_ring_kernel<true>), grid, dim3(256), lds, st, p);
                else hipLaunchKernelGGL((attn_d72_ring_kernel<false>), grid, dim3(256), lds, st, p);
                return hipGetLastError();
            }
        }
        if (a.d > 64 && a.d <= 80) return launch_rowmajor<3, 5, true>(p, a, st, f16);          // SigLIP-so400m: 72 -> five 16-wide output tiles, known at compile time
        if (a.d <= 96) return launch_rowmajor<3, 6>(p, a, st, f16);
        return launch_rowmajor<4, 8>(p, a, st, f16);
    }
    if (variant == 2 && !can_mfma) return hipErrorInvalidValue;
    if (variant == 6) {          // the chunk kernel's contiguous decomposition, forced (attn_chunk.h)
        if (!can_gqa128 || a.qkv_slabs || a.dyn || a.S * (a.nh / a.nkv) < 256) return hipErrorInvalidValue;
        g_last_form[0] = 8;
        return launch_gqa128_chunk(p, a, st);
    }
    if (variant == 5) {          // software-pipelined waves, balanced row blocks (attn_w1.h)
        if (!can_gqa128 || a.qkv_slabs || a.dyn) return hipErrorInvalidValue;
        g_last_form[0] = 5;
        return launch_gqa128_w1_auto(p, a, st);
    }
    if (variant == 3) {
        if (!can_gqa128) return hipErrorInvalidValue;
        const int rows_total = a.S * (a.nh / a.nkv);
        if (a.qkv_slabs && (rows_total > 64 || a.k_ts != 128 || a.n_slabs < 1 || a.n_slabs > 4)) return hipErrorInvalidValue;        // the fused q/k/v preparation lives in the decode form only
        // per-frame steps and short chunks over a long context (17 .. 768 stacked rows per kv head, >= 64 key tiles): attn_gqa128_w1_kernel -- two balanced row blocks of whole
        // 16-row tiles instead of three 128-row blocks, one block per CU on the four-slot ring (S = 49 over 15 k keys: 31.6 -> 27.7 us incl. the merge; below 4 k keys its
        // longer pipeline fill loses 2-20 %, from 1 k rows up the 256-row phase-split form is 8 % faster: profiles/r05_attn_small_s.md)
        if (a.variant == 0 && !a.qkv_slabs && !a.dyn && rows_total > 16 && rows_total <= 768 && a.n_ctx + a.S >= 4096 && a.ws) { g_last_form[0] = 5; return launch_gqa128_w1_auto(p, a, st); }
        g_last_form[0] = rows_total <= 64 ? 3 : 4;
        return rows_total <= 64 ? launch_gqa128<1>(p, a, st) : launch_gqa128<2>(p, a, st);
    }
    g_last_form[0] = variant == 1 ? 1 : 2;
    if (variant == 1) {
        if (a.d > 128) return hipErrorInvalidValue;
        dim3 grid(a.S, a.nh, a.batch);
        if (dtype == MMD_F32) hipLaunchKernelGGL(attn_simple_kernel<float>, grid, dim3(64), 0, st, p);
        else hipLaunchKernelGGL(attn_simple_kernel<bf16_t>, grid, dim3(64), 0, st, p);
        return hipGetLastError();
    }
    if (a.d <= 32) return launch_mfma<32>(p, a, st);
    if (a.d <= 64) return launch_mfma<64>(p, a, st);
    if (a.d <= 96) return launch_mfma<96>(p, a, st);
    return launch_mfma<128>(p, a, st);
}
