// gemm.hip -- Y[M,N] = epilogue(X[M,K] . W[N,K]^T + bias) on MFMA (gfx950), bf16 and fp32.
//
// Stands in for every nn.Linear of the path: Qwen2 q/k/v/o/gate/up/down (transformers qwen2/modeling_qwen2.py:36-48,
// 200-240), lm_head / informative_head / relevance_head (models/live_llava/video_head_live_llava_qwen.py:76-78), SigLIP
// q/k/v/out/fc1/fc2 + the patch-embed conv as im2col GEMM (siglip/modeling_siglip.py:124-186, 250-322) and the
// mm_projector.  W keeps the checkpoint's [out,in] layout, so both operands are K-contiguous ("B^T input").
//
// Operand order is swapped (acc = mfma(W_frag, X_frag)): a lane then owns 4 CONSECUTIVE n for one m, so the epilogue
// reads bias/residual and writes the result as one 8-byte (bf16) / 16-byte (fp32) access per 16x16 tile.
//
// Rounding points follow eager bf16 execution of the reference: linear output rounded to the storage type before the
// activation / residual add / gate*up product (no-ops in fp32).
#include "common.h"
#include <type_traits>

#include "gemm_ring.h"

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { using type = bf16x8_t; static constexpr int KSTEP = 32; };
template <> struct Frag<float> { using type = float; static constexpr int KSTEP = 4; };

template <typename T, int BM, int BN> struct TileCfg {
    static constexpr int BK = 32;
    static constexpr int PAD = sizeof(T) == 2 ? 8 : 1;
    static constexpr int LD = BK + PAD;
    static constexpr int WM = BM / 2, WN = BN / 2;      // 2x2 waves
    static constexpr int TM = WM / 16, TN = WN / 16;
};

// cooperative, bounds-checked load of a [ROWS][32] tile (rows r0.., cols k0..) into LDS (zero-filled outside)
template <typename T, int ROWS, int LD>
__device__ __forceinline__ void stage_tile(T* __restrict__ lds, const T* __restrict__ g, long long ld, int r0, int nrows, int k0, int kend,
                                           int vec, int tid) {
    if constexpr (sizeof(T) == 2) {
        // 4 chunks of 8 elements per row
        for (int i = tid; i < ROWS * 4; i += 256) {
            int r = i >> 2, c = (i & 3) * 8;
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            int gr = r0 + r, gk = k0 + c;
            if (gr < nrows && gk < kend) {
                const T* src = g + (long long)gr * ld + gk;
                if (vec && gk + 8 <= kend) {
                    v = *reinterpret_cast<const s16x8_t*>(src);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (gk + e < kend) v[e] = (short)src[e];
                }
            }
            *reinterpret_cast<s16x8_t*>(lds + r * LD + c) = v;
        }
    } else {
        for (int i = tid; i < ROWS * 32; i += 256) {
            int r = i >> 5, c = i & 31;
            int gr = r0 + r, gk = k0 + c;
            T v = 0;
            if (gr < nrows && gk < kend) v = g[(long long)gr * ld + gk];
            lds[r * LD + c] = v;
        }
    }
}

template <typename T>
__device__ __forceinline__ void store4(const GemmP& p, int m, int n, const float (&v)[4]) {
    // v = acc (+bias) for columns n..n+3 of row m, epilogues except SWIGLU
    if (m >= p.M) return;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = v[r];
        if (n + r < p.N && p.wscale) x *= p.wscale[n + r];
        if (n + r < p.N && p.bias) x += to_f<T>(((const T*)p.bias)[n + r]);
        if (p.epi == EPI_GELU_TANH) x = gelu_tanh_t<T>(rnd<T>(x));
        else if (p.epi == EPI_GELU_ERF) x = gelu_erf_t<T>(rnd<T>(x));
        else if (p.epi == EPI_RESID) { if (n + r < p.N) x = rnd<T>(x) + to_f<T>(((const T*)p.R)[(long long)m * p.ldr + n + r]); }
        o[r] = x;
    }
    if (p.out_f32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = rnd<T>(o[r]);
        float* y = (float*)p.Y + (long long)m * p.ldy + n;
        if (n + 3 < p.N && ((p.ldy & 3) == 0)) *reinterpret_cast<f32x4_t*>(y) = f32x4_t{o[0], o[1], o[2], o[3]};
        else { for (int r = 0; r < 4; ++r) if (n + r < p.N) y[r] = o[r]; }
    } else {
        T* y = (T*)p.Y + (long long)m * p.ldy + n;
        if (n + 3 < p.N && ((p.ldy & 3) == 0)) {
            if constexpr (sizeof(T) == 2) {
                s16x4_t pk = {(short)f2bf(o[0]), (short)f2bf(o[1]), (short)f2bf(o[2]), (short)f2bf(o[3])};
                *reinterpret_cast<s16x4_t*>(y) = pk;
            } else {
                *reinterpret_cast<f32x4_t*>(y) = f32x4_t{o[0], o[1], o[2], o[3]};
            }
        } else { for (int r = 0; r < 4; ++r) if (n + r < p.N) y[r] = from_f<T>(o[r]); }
    }
}

template <typename T>
__device__ __forceinline__ void store4_swiglu(const GemmP& p, int m, int n_gate, const float (&g)[4], const float (&u)[4]) {
    // interleaved layout: columns [32j, 32j+16) = gate rows 16j.., [32j+16, 32j+32) = up rows 16j..
    if (m >= p.M) return;
    int oc = (n_gate >> 5) * 16 + (n_gate & 15);
    int NO = p.N >> 1;
    T* y = (T*)p.Y + (long long)m * p.ldy + oc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (oc + r < NO) {
            float gs = g[r], us = u[r];
            if (p.wscale) { gs *= p.wscale[n_gate + r]; us *= p.wscale[n_gate + 16 + r]; }
            float gg = rnd<T>(gs), uu = rnd<T>(us);
            float s = rnd<T>(silu_t<T>(gg));
            y[r] = from_f<T>(s * uu);
        }
    }
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmP p) {
    using C = TileCfg<T, BM, BN>;
    using F = Frag<T>;
    __shared__ __attribute__((aligned(16))) T Xs[BM * C::LD];
    __shared__ __attribute__((aligned(16))) T Ws[BN * C::LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * C::WM, wn = (wave & 1) * C::WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const int lr = lane & 15, lq = lane >> 4;

    f32x4_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

    for (int k0 = kbeg; k0 < kend; k0 += C::BK) {
        __syncthreads();
        stage_tile<T, BM, C::LD>(Xs, (const T*)p.X, p.ldx, m0, p.M, k0, kend, p.vec, tid);
        stage_tile<T, BN, C::LD>(Ws, (const T*)p.W, p.ldw, n0, p.N, k0, kend, p.vec, tid);
        __syncthreads();
        if constexpr (sizeof(T) == 2) {
            bf16x8_t xf[C::TM], wf[C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(Xs + (wm + i * 16 + lr) * C::LD + lq * 8);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(Ws + (wn + j * 16 + lr) * C::LD + lq * 8);
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < C::BK; kk += 4) {
                float xf[C::TM], wf[C::TN];
#pragma unroll
                for (int i = 0; i < C::TM; ++i) xf[i] = Xs[(wm + i * 16 + lr) * C::LD + kk + lq];
#pragma unroll
                for (int j = 0; j < C::TN; ++j) wf[j] = Ws[(wn + j * 16 + lr) * C::LD + kk + lq];
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // epilogue: lane holds C[m = .. + lr][n = .. + lq*4 + r]
    if (gridDim.z > 1) {
        float* ws = p.ws + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                int m = m0 + wm + i * 16 + lr, n = n0 + wn + j * 16 + lq * 4;
                if (m < p.M) {
                    float* y = ws + (long long)m * p.N + n;
                    if (n + 3 < p.N && (p.N & 3) == 0) *reinterpret_cast<f32x4_t*>(y) = acc[i][j];
                    else for (int r = 0; r < 4; ++r) if (n + r < p.N) y[r] = acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
        int m = m0 + wm + i * 16 + lr;
        if (p.epi == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < C::TN; j += 2) {
                int n = n0 + wn + j * 16 + lq * 4;
                float g[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                float u[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                store4_swiglu<T>(p, m, n, g, u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                int n = n0 + wn + j * 16 + lq * 4;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                store4<T>(p, m, n, v);
            }
        }
    }
}

// sums split-K slabs and applies the epilogue; one thread per 4 consecutive columns
template <typename T>
__global__ void splitk_reduce_kernel(GemmP p, int splits) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int n4 = (p.N + 3) / 4;
    if (idx >= (long long)p.M * n4) return;
    int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
    if (p.epi == EPI_SWIGLU) {
        // handle a gate quad; its up partner sits 16 columns further
        if ((n & 16) != 0) return;
        float g[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
        for (int s = 0; s < splits; ++s) {
            const float* w = p.ws + ((long long)s * p.M + m) * p.N + n;
            for (int r = 0; r < 4; ++r) { if (n + r < p.N) g[r] += w[r]; if (n + 16 + r < p.N) u[r] += w[16 + r]; }
        }
        store4_swiglu<T>(p, m, n, g, u);
        return;
    }
    float v[4] = {0, 0, 0, 0};
    for (int s = 0; s < splits; ++s) {
        const float* w = p.ws + ((long long)s * p.M + m) * p.N + n;
        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += w[r];
    }
    store4<T>(p, m, n, v);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-streaming GEMM for M <= 16*MT rows (the per-frame LLM step, M ~ 49, and M = 1 decoding): HBM-bound, every
// weight byte is read exactly once.
//   * W is stored MFMA-fragment-major ("packed", pack_w16x32_kernel): tile (nt, kt) = 16 rows x 32 k is one contiguous
//     1 KiB block whose 16-byte piece `lane` is exactly that lane's A operand.  A wave streams its n-tile along K with
//     one fully coalesced 1 KiB non-temporal load per MFMA k-step, straight into VGPRs (no LDS round trip for the
//     operand that is used once).
//   * X [M,K] (L2-resident, re-read by every block) is staged through LDS once per block in full 256-byte row pieces
//     and shared by the 4 waves; the LDS image is fragment-ordered with an XOR on the row index so that both the
//     row-contiguous ds_write_b128 and the fragment ds_read_b128 are bank-conflict-free.
//   * 4 waves = 4*NT n-tiles per block; K is split over grid.y into fp32 slabs when N alone cannot fill 256 CUs
//     (deterministic slab reduce, no atomics).
// ------------------------------------------------------------------------------------------------------------------
__global__ void pack_w16x32_kernel(const bf16_t* __restrict__ W, long long ldw, int N, int K, bf16_t* __restrict__ out) {
    // one thread per 16-byte piece: out[((nt*KT + kt)*64 + lane)*8 + e] = W[nt*16 + (lane&15)][kt*32 + (lane>>4)*8 + e]
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int KT = K >> 5;
    long long pieces = (long long)(N >> 4) * KT * 64;
    if (i >= pieces) return;
    int lane = (int)(i & 63);
    long long tile = i >> 6;
    int kt = (int)(tile % KT);
    long long nt = tile / KT;
    const bf16_t* src = W + (nt * 16 + (lane & 15)) * ldw + kt * 32 + (lane >> 4) * 8;
    *reinterpret_cast<s16x8_t*>(out + i * 8) = *reinterpret_cast<const s16x8_t*>(src);
}

hipError_t launch_pack_w(const void* W, int64_t ldw, int N, int K, void* out, hipStream_t st) {
    long long pieces = (long long)(N >> 4) * (K >> 5) * 64;
    if (pieces <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_w16x32_kernel, dim3(cdiv(pieces, 256)), dim3(256), 0, st, (const bf16_t*)W, (long long)ldw, N, K, (bf16_t*)out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// fp8 e4m3 (OCP e4m3fn) weight path (BASELINE configs[4]: "fp8 MFMA weights"; SURVEY.md section 8d row 5: per-channel-scaled
// weights, bf16 activations, fp32 accumulate).  gfx950's plain fp8 MFMA runs at the bf16 rate and would need fp8 ACTIVATIONS, so the
// gain is bytes, not FLOPs: the weight-streaming kernels (M <= 64: decode GEMV, per-frame steps) read the 1-byte copy and widen it to
// bf16 in registers (v_cvt_scalef32_pk_bf16_fp8, exact), the MFMA-bound tile kernels keep reading a bf16 copy of the SAME values q.
// Every regime computes (sum_k x_k q_nk) * scale_n with exact products and fp32 accumulation, so the chunked and the per-frame schedule
// still agree up to accumulation order.  scale_n = amax_n / 448, q = round-to-nearest-even(W / scale_n) (v_cvt_pk_fp8_f32).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(bf16_t* __restrict__ W, int K, uint8_t* __restrict__ q8, float* __restrict__ scale) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    bf16_t* w = W + row * K;
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(bf2f(w[k])));
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scale[row] = sc;
    for (int k = threadIdx.x * 2; k < K; k += 512) {                // K is even
        const float a = bf2f(w[k]) / sc, b = bf2f(w[k + 1]) / sc;   // true division: the host reference does W / scale
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
        q8[row * K + k] = (uint8_t)(pk & 0xff); q8[row * K + k + 1] = (uint8_t)((pk >> 8) & 0xff);
        const auto back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
        w[k] = f2bf(back[0]); w[k + 1] = f2bf(back[1]);             // e4m3 has 3 mantissa bits: exact in bf16
    }
}
hipError_t launch_quantize_fp8_rows(void* W, int N, int K, uint8_t* q8, float* scale, hipStream_t st) {
    if (N <= 0 || (K & 1)) return N <= 0 ? hipSuccess : hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3(N), dim3(256), 0, st, (bf16_t*)W, K, q8, scale);
    return hipGetLastError();
}
// fragment-major fp8 layout: block (nt, kp) = 16 rows x 64 k = 1 KiB; lane l's 16 bytes = row (l & 15), k = kp*64 + (l >> 4)*8 + {0..7} (the
// A operand of k-step 2kp) followed by k = kp*64 + 32 + (l >> 4)*8 + {0..7} (k-step 2kp + 1): one coalesced 1 KiB load feeds two MFMAs
__global__ void pack_w8_kernel(const uint8_t* __restrict__ q8, int N, int K, uint8_t* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KP = K >> 6;
    const long long pieces = (long long)(N >> 4) * KP * 64;
    if (i >= pieces) return;
    const int lane = (int)(i & 63);
    const long long blk = i >> 6;
    const int kp = (int)(blk % KP);
    const long long nt = blk / KP;
    const uint8_t* src = q8 + (nt * 16 + (lane & 15)) * (long long)K + kp * 64 + (lane >> 4) * 8;
    uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 32);
    *reinterpret_cast<uint4*>(out + i * 16) = uint4{lo.x, lo.y, hi.x, hi.y};
}
hipError_t launch_pack_w8(const uint8_t* q8, int N, int K, void* out, hipStream_t st) {
    const long long pieces = (long long)(N >> 4) * (K >> 6) * 64;
    if (pieces <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_w8_kernel, dim3(cdiv(pieces, 256)), dim3(256), 0, st, q8, N, K, (uint8_t*)out);
    return hipGetLastError();
}
// 8 fp8 (two dwords) -> 8 bf16, exact
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16(unsigned lo, unsigned hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)lo, 1.0f, true);
    const b2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)hi, 1.0f, true);
    return bf16x8_t{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

__device__ __forceinline__ int xs_slot(int ktl, int kq, int r) { return kq * 16 + (r ^ (kq + 4 * (ktl & 1))); }

template <int MT, int NT, bool W8 = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmP p, int KT, int kt_per_split) {
    constexpr int KS = 4;                                   // k-tiles (of 32) per pipeline step
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][MT * KS * 64 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int ntiles = p.N >> 4;
    const int nt0 = (blockIdx.x * 4 + wave) * NT;
    const int kt_beg = blockIdx.y * kt_per_split;
    const int kt_end = min(KT, kt_beg + kt_per_split);
    const bf16_t* X = (const bf16_t*)p.X;
    const bf16_t* Wp = (const bf16_t*)p.W;

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

    bf16x8_t wc[NT][KS], wn[NT][KS];
    s16x8_t xr[MT];

    auto load_w = [&](bf16x8_t (&w)[NT][KS], int kt) {
        if constexpr (W8) {
            // fp8 copy: one 16-byte piece per lane = the operands of two k-steps (kt is a multiple of 4 here, K % 64 == 0)
            const uint8_t* W8p = (const uint8_t*)p.W;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int s = 0; s < KS; s += 2) {
                    u32x4_t v = {0, 0, 0, 0};
                    if (nt0 + j < ntiles && kt + s < kt_end)
                        v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(W8p + (((long long)(nt0 + j) * (KT >> 1) + ((kt + s) >> 1)) * 64 + lane) * 16));
                    w[j][s] = fp8x8_to_bf16(v[0], v[1]); w[j][s + 1] = fp8x8_to_bf16(v[2], v[3]);
                }
            return;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (nt0 + j < ntiles && kt + s < kt_end)
                    v = __builtin_nontemporal_load(reinterpret_cast<const s16x8_t*>(Wp + (((long long)(nt0 + j) * KT + kt + s) * 64 + lane) * 8));
                w[j][s] = __builtin_bit_cast(bf16x8_t, v);
            }
    };
    auto load_x = [&](int kt) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            int i = tid + 256 * j, m = i >> 4, c = i & 15;
            s16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M && kt + (c >> 2) < kt_end) v = *reinterpret_cast<const s16x8_t*>(X + (long long)m * p.ldx + kt * 32 + c * 8);
            xr[j] = v;
        }
    };
    auto write_x = [&](int buf) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            int i = tid + 256 * j, m = i >> 4, c = i & 15;
            int mt = m >> 4, r = m & 15, ktl = c >> 2, kq = c & 3;
            *reinterpret_cast<s16x8_t*>(&Xs[buf][((mt * KS + ktl) * 64 + xs_slot(ktl, kq, r)) * 8]) = xr[j];
        }
    };

    load_w(wc, kt_beg);
    load_x(kt_beg);
    write_x(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt_beg; kt < kt_end; kt += KS) {
        const bool more = kt + KS < kt_end;
        if (more) { load_w(wn, kt + KS); load_x(kt + KS); }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (kt + s < kt_end) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(&Xs[cur][((i * KS + s) * 64 + xs_slot(s, lq, lr)) * 8]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[j][s], xf, acc[i][j], 0, 0, 0);
                }
            }
        }
        if (more) write_x(cur ^ 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) wc[j][s] = wn[j][s];
        cur ^= 1;
    }

    if (gridDim.y > 1 || p.slabs) {
        float* ws = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int m = i * 16 + lr, n = (nt0 + j) * 16 + lq * 4;
                if (m < p.M && nt0 + j < ntiles) {
                    f32x4_t v = acc[i][j];
                    if (p.slabs && p.wscale) v *= *reinterpret_cast<const f32x4_t*>(p.wscale + n);      // the fused slab consumers know no scale (split-K reduce applies it itself)
                    *reinterpret_cast<f32x4_t*>(ws + (long long)m * p.N + n) = v;
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = i * 16 + lr;
        if (p.epi == EPI_SWIGLU) {
            if constexpr (NT >= 2) {
#pragma unroll
                for (int j = 0; j < NT; j += 2) {
                    if (nt0 + j + 1 < ntiles) {
                        int n = (nt0 + j) * 16 + lq * 4;
                        float g[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        float u[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                        store4_swiglu<bf16_t>(p, m, n, g, u);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (nt0 + j < ntiles) {
                    int n = (nt0 + j) * 16 + lq * 4;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store4<bf16_t>(p, m, n, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// M <= 16 (decode, M = 1): pure weight streaming.  A block owns NT n-tiles; its 4 waves split K four ways and each
// wave streams its part of the packed W with U independent 1 KiB non-temporal loads in flight (no LDS, no barrier in
// the loop -- "GEMV: load straight to VGPRs, deep unroll, late wait").  X rows (<= 16, L2-resident) are read as MFMA
// operands directly.  The four partial accumulators meet in LDS once; wave 0 applies the epilogue (or leaves ONE fp32
// slab for the fused consumer): no split-K slabs, no reduce launch.
// ------------------------------------------------------------------------------------------------------------------
// x fragment of the decode chain: rnd(gamma * rnd(h * inv)) (Qwen2RMSNorm's rounding points)
__device__ __forceinline__ s16x8_t chain_norm8(const s16x8_t& h, const s16x8_t& g, float inv) {
    s16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bf2f((bf16_t)g[e]) * bf2f(f2bf(bf2f((bf16_t)h[e]) * inv)));
    return o;
}

template <int NT, int U, bool W8 = false, bool CHAIN = false, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void gemm_gemv16_kernel(GemmP p, int KT, GemvChain ch) {
    __shared__ __attribute__((aligned(16))) float red[WAVES - 1][NT][64][4];
    // consumer side of the decode chain: each wave keeps the normalised activation of ITS K range here (rows <= GEMV_CHAIN_ROWS, <= 1024 k per wave)
    constexpr int XW = 1024;
    __shared__ __attribute__((aligned(16))) bf16_t xs[(CHAIN && WAVES == 4) ? WAVES * GEMV_CHAIN_ROWS * XW : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.x * NT;
    // K range of this block (grid.y splits K into slabs when N alone gives too few blocks), then of this wave; in k-tiles of 32 (fp8: pairs of tiles)
    constexpr int KG = W8 ? 2 : 1;
    const int KTg = KT / KG;
    const int ktb = (KTg + gridDim.y - 1) / gridDim.y;
    const int blk_beg = blockIdx.y * ktb, blk_end = min(KTg, blk_beg + ktb);
    const int ktw = (blk_end - blk_beg + WAVES - 1) / WAVES;
    const int kt_beg = blk_beg + wave * ktw, kt_end = min(blk_end, kt_beg + ktw);      // units of KG k-tiles
    const bool xn = CHAIN && WAVES == 4 && ch.xn_h != nullptr;        // X built from the residual stream (see GemvChain)
    const bf16_t* xrow = (const bf16_t*)p.X + (long long)lr * p.ldx + lq * 8;
    const bf16_t* xl = xs + (wave * GEMV_CHAIN_ROWS + lr) * XW + lq * 8 - kt_beg * (32 * KG);     // + k: this wave's copy of row lr
    const bool row_ok = lr < p.M;

    f32x4_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4_t{0, 0, 0, 0};

    // chain prologue, run by each wave AFTER its first weight loads are issued: 1/rms of every row from the producer's per-n-tile partial sums
    // (lane l takes entries 4l..4l+3; fixed order), then rnd(gamma * rnd(h * inv)) of the wave's K range into LDS
    auto chain_prologue = [&]() {
        const int k0 = kt_beg * (32 * KG), kn = (kt_end - kt_beg) * (32 * KG);
        for (int r = 0; r < p.M; ++r) {
            const f32x4_t q = *reinterpret_cast<const f32x4_t*>(ch.xn_ssq + r * GEMV_SSQ_STRIDE + lane * 4);
            s16x8_t hv[2], gv[2];
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int k = ps * 512 + lane * 8;
                hv[ps] = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; gv[ps] = hv[ps];
                if (k < kn) {
                    hv[ps] = *reinterpret_cast<const s16x8_t*>((const bf16_t*)ch.xn_h + (long long)r * p.K + k0 + k);
                    gv[ps] = *reinterpret_cast<const s16x8_t*>((const bf16_t*)ch.xn_gamma + k0 + k);
                }
            }
            const float inv = rsqrtf(wave_sum((q[0] + q[1]) + (q[2] + q[3])) / (float)p.K + ch.xn_eps);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int k = ps * 512 + lane * 8;
                if (k < kn) *reinterpret_cast<s16x8_t*>(xs + (wave * GEMV_CHAIN_ROWS + r) * XW + k) = chain_norm8(hv[ps], gv[ps], inv);
            }
        }
    };

    if constexpr (W8) {
        // fp8 weights: K ranges are in 64-k pairs; a 1 KiB load feeds two MFMAs; U/2 loads per n-tile in flight
        const uint8_t* W8p = (const uint8_t*)p.W;
        const int KP = KT >> 1;
        constexpr int UP = U / 2;
        auto iter = [&](int kp0, auto first) {
            u32x4_t wv[NT][UP]; bf16x8_t x[UP][2];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int kp = kp0 + u;
                s16x8_t x0 = {0, 0, 0, 0, 0, 0, 0, 0}, x1 = x0;
                if (kp < kt_end) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        wv[j][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(W8p + (((long long)(nt0 + j) * KP + kp) * 64 + lane) * 16));
                    if (!xn && row_ok) { x0 = *reinterpret_cast<const s16x8_t*>(xrow + kp * 64); x1 = *reinterpret_cast<const s16x8_t*>(xrow + kp * 64 + 32); }
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) wv[j][u] = u32x4_t{0, 0, 0, 0};
                }
                x[u][0] = __builtin_bit_cast(bf16x8_t, x0); x[u][1] = __builtin_bit_cast(bf16x8_t, x1);
            }
            if constexpr (CHAIN) if (xn) {
                if constexpr (decltype(first)::value) chain_prologue();
#pragma unroll
                for (int u = 0; u < UP; ++u) if (row_ok && kp0 + u < kt_end) {
                    x[u][0] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const s16x8_t*>(xl + (kp0 + u) * 64));
                    x[u][1] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const s16x8_t*>(xl + (kp0 + u) * 64 + 32));
                }
            }
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16(wv[j][u][0], wv[j][u][1]), x[u][0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16(wv[j][u][2], wv[j][u][3]), x[u][1], acc[j], 0, 0, 0);
                }
        };
        if (kt_beg < kt_end) {
            iter(kt_beg, std::true_type{});
            for (int kp0 = kt_beg + UP; kp0 < kt_end; kp0 += UP) iter(kp0, std::false_type{});
        }
    } else {
        const bf16_t* Wp = (const bf16_t*)p.W;
        auto iter = [&](int kt0, auto first) {
            bf16x8_t w[NT][U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kt = kt0 + u;
                s16x8_t xv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kt < kt_end) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        w[j][u] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load(reinterpret_cast<const s16x8_t*>(Wp + (((long long)(nt0 + j) * KT + kt) * 64 + lane) * 8)));
                    if (!xn && row_ok) xv = *reinterpret_cast<const s16x8_t*>(xrow + kt * 32);
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) w[j][u] = __builtin_bit_cast(bf16x8_t, xv);
                }
                x[u] = __builtin_bit_cast(bf16x8_t, xv);
            }
            if constexpr (CHAIN) if (xn) {
                if constexpr (decltype(first)::value) chain_prologue();
#pragma unroll
                for (int u = 0; u < U; ++u) if (row_ok && kt0 + u < kt_end) x[u] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const s16x8_t*>(xl + (kt0 + u) * 32));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][u], x[u], acc[j], 0, 0, 0);
        };
        if (kt_beg < kt_end) {
            iter(kt_beg, std::true_type{});
            for (int kt0 = kt_beg + U; kt0 < kt_end; kt0 += U) iter(kt0, std::false_type{});
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4_t*>(&red[wave - 1][j][lane][0]) = acc[j];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int w2 = 0; w2 < WAVES - 1; ++w2) acc[j] += *reinterpret_cast<const f32x4_t*>(&red[w2][j][lane][0]);
    const int m = lr;
    if constexpr (CHAIN) if (ch.fin_h) {
        // producer side of the decode chain: the block owns its n-tile over ALL of K (grid.y == 1; the K split is over the block's waves), so the
        // residual add and the tile's sum of squares need no other block -- cross-block hand-over inside a kernel costs an L2 write-back on this part
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float ss = 0.f;
            if (m < p.M) {
                f32x4_t v = acc[j];
                if (p.wscale) v *= *reinterpret_cast<const f32x4_t*>(p.wscale + (nt0 + j) * 16 + lq * 4);
                bf16_t* hp = (bf16_t*)ch.fin_h + (long long)m * p.N + (nt0 + j) * 16 + lq * 4;
                const s16x4_t r = *reinterpret_cast<const s16x4_t*>(hp);
                s16x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hv = bf2f(f2bf(bf2f(f2bf(v[e])) + bf2f((bf16_t)r[e])));      // rnd(rnd(gemm) + residual)
                    ss += hv * hv; o[e] = (short)f2bf(hv);
                }
                *reinterpret_cast<s16x4_t*>(hp) = o;
            }
            ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
            if (lq == 0 && m < p.M) ch.fin_ssq[m * GEMV_SSQ_STRIDE + nt0 + j] = ss;
        }
        return;
    }
    if (m >= p.M) return;
    if (p.slabs) {
        float* ws = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4_t v = acc[j];
            if (p.wscale) v *= *reinterpret_cast<const f32x4_t*>(p.wscale + (nt0 + j) * 16 + lq * 4);       // slabs feed the fused consumers, which know no scale
            *reinterpret_cast<f32x4_t*>(ws + (long long)m * p.N + (nt0 + j) * 16 + lq * 4) = v;
        }
        return;
    }
    if (p.epi == EPI_SWIGLU) {
        if constexpr (NT >= 2) {
#pragma unroll
            for (int j = 0; j < NT; j += 2) {
                float g[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                float u[4] = {acc[j + 1][0], acc[j + 1][1], acc[j + 1][2], acc[j + 1][3]};
                store4_swiglu<bf16_t>(p, m, (nt0 + j) * 16 + lq * 4, g, u);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
            store4<bf16_t>(p, m, (nt0 + j) * 16 + lq * 4, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_stream_kernel<MT, NT, WK, KS, NB>: the weight-streaming GEMM of the 16 < M <= 256 regime (round 4) -- the reference's own
// per-frame schedule (M = 49 + prefix, test/inference.py:221-246), the demo path (demo/liveinfer.py:69-105), short chunks and the remainders behind a response.
// Up to M = 256 a forward is bound by the 466 MB of weights a layer streams (16 MFMAs per 1 KB weight tile fill the CU's four SIMDs only at M = 256), yet
// gemm_skinny_kernel reads 3.3-4.6 TB/s at M = 49 and 64 < M <= 256 fell to the 128-row tile kernel (2.4-3.0 x the weight-stream time).  Measured causes
// (profiles/r04_step49_kernels_before.txt): X staged through registers behind the weight loads (waiting for X = vmcnt(0): nothing stays in flight across the step
// barrier), one step of lookahead, and 296 four-wave blocks for gate_up -- most CUs hold 4 waves x 8 KB in flight, below what 25 GB/s per CU needs.
//
//   * block = 4 x WK waves: wave (wn, wk) owns NT n-tiles (16 columns each) and the k-tiles wk * KS .. of every step (the block's K range advances KSB = WK * KS
//     k-tiles per step); all MT m-tiles (M <= 16 MT rows).  The WK partial sums meet in LDS once at the end.
//   * W: packed fragment tiles straight into registers by inline-asm non-temporal loads, LA = NB - 1 steps ahead through NB register buffers (NT * KS KB per wave
//     and step): LA * NT * KS KB in flight per wave at all times.
//   * X: the step's [16 MT rows][32 KSB k] slab by LDS-DMA into an NB-slot ring, in the ring GEMM's piece format (16 rows x 64 B pieces, 16-byte chunk XOR-swizzled:
//     conflict-free b128 fragment reads), issued LA steps ahead like W; every wave reads the fragments of its own k-tiles at use.
//   * ONE hand-counted `s_waitcnt vmcnt((LA - 1) * (NT * KS + XP))` + raw s_barrier per step (W (s) and X (s) were both issued in step s - LA; the LA - 1 steps behind
//     them stay in flight); hipcc sees none of the W loads (inline asm) and never waits for them.
//   * epilogue: fp32 slabs for the fused consumers (reduce + bias + RoPE + append / reduce + residual + RMSNorm), or the SwiGLU / plain epilogues in place.
// ------------------------------------------------------------------------------------------------------------------
template <int N_, typename F> __device__ __forceinline__ void stream_static_for(F&& f) {
    if constexpr (N_ > 0) { stream_static_for<N_ - 1>(f); f(std::integral_constant<int, N_ - 1>{}); }
}

template <int MT, int NT, int WK, int KS, int NB, int DBG = 0, int WN = 4>          // DBG (timing only, wrong results): 1 = W stream alone (no X DMA, no MFMA), 2 = no W loads
__global__ __launch_bounds__(64 * WN * WK) void gemm_stream_kernel(GemmP p, int KT, int kt_per_block) {
    constexpr int NWAVE = WN * WK, KSB = WK * KS;
    constexpr int XPIECES = MT * KSB;                        // 1 KB pieces of the step's X slab
    constexpr int XP = DBG == 1 ? 0 : (XPIECES + NWAVE - 1) / NWAVE;       // LDS-DMAs per wave and step: the SAME count for every wave (the hand-counted waits need that): where the
                                                                           // pieces do not divide evenly, the surplus DMAs re-load pieces 0.. (same bytes to the same place)
    constexpr int NWL = DBG == 2 ? 0 : NT * KS;              // W loads per wave and step
    constexpr int LA = NB - 1;
    constexpr int ISSUE = NWL + XP;
    constexpr int VM_STEP = (LA - 1) * ISSUE;
    constexpr int XSLOT = XPIECES * 512;                     // elements per ring slot
    static_assert(VM_STEP <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wn = wave % WN, wk = wave / WN;
    const int ntiles = p.N >> 4;
    const int nt0 = (blockIdx.x * WN + wn) * NT;          // (a wave whose n-tiles lie beyond N streams the last tile again and stores nothing)
    const int kt_beg = blockIdx.y * kt_per_block;
    const int kt_end = min(KT, kt_beg + kt_per_block);
    const int nsteps = (kt_end - kt_beg) / KSB;              // (dispatch condition: every block's K range is a whole number of steps)
    const bf16_t* X = (const bf16_t*)p.X;
    const bf16_t* Wp = (const bf16_t*)p.W;

    // X piece pi = (m-tile pi / KSB, k-tile pi % KSB of the step): lane -> row srow, 16-byte chunk spos ^ sswz (the ring GEMM's piece image)
    const int srow = lane >> 2, spos = lane & 3;
    const int sswz = (0x1230 >> (((srow >> 2) & 3) * 4)) & 3;
    unsigned xo[XP + 1];
#pragma unroll
    for (int j = 0; j < XP; ++j) {
        const int pi = (wave + NWAVE * j) % XPIECES;
        int row = (pi / KSB) * 16 + srow; row = row < p.M ? row : p.M - 1;
        xo[j] = (unsigned)(row * (int)p.ldx + (pi % KSB) * 32 + ((spos ^ sswz) * 8)) * 2u;
    }
    long long wo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { int nt = nt0 + j; nt = nt < ntiles ? nt : ntiles - 1; wo[j] = ((long long)nt * KT + kt_beg + wk * KS) * 1024; }
    const unsigned wlane = lane * 16;
    auto dma_x = [&](int slot, int step, int j) {
        const char* ub = (const char*)X + ((long long)kt_beg + (long long)step * KSB) * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + xo[j]),
                                         (__attribute__((address_space(3))) void*)(lds + slot * XSLOT + ((wave + NWAVE * j) % XPIECES) * 512), 16, 0, 0);
    };
    auto load_w = [&](bf16x8_t& dst, int j, int ks, int step) {
        const unsigned long long ua = (unsigned long long)(uintptr_t)((const char*)Wp + wo[j] + ((long long)step * KSB + ks) * 1024);
        // (under SGPR pressure hipcc parks uniform values in VGPR lanes; an asm "s" operand cannot take them back -- readfirstlane can)
        const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ua >> 32)) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ua);          // (the builtin returns int: widen as UNSIGNED)
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(wlane), "s"(ub) : "memory");
    };
    const int rswz = (0x1230 >> (((lr >> 2) & 3) * 4)) & 3;
    const int aoff = lr * 32 + ((lq ^ rswz) * 8);

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};
    bf16x8_t wb[NB][NT][KS];

    auto issue = [&](int step, auto buf_c) {          // everything step `step` needs, into register buffer / ring slot `buf`
        constexpr int buf = decltype(buf_c)::value;
        if constexpr (DBG != 2) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) load_w(wb[buf][j][ks], j, ks, step);
        }
#pragma unroll
        for (int j = 0; j < XP; ++j) dma_x(buf, step, j);
    };
    // prologue: steps 0 .. LA - 1
    stream_static_for<LA>([&](auto uc) { constexpr int u = decltype(uc)::value; if (u < nsteps) issue(u, std::integral_constant<int, u>{}); });

    auto step = [&](int s, auto buf_c, bool steady) {
        constexpr int cur = decltype(buf_c)::value, nxt = (cur + LA) % NB;
        if (steady) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(VM_STEP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tail: the steps behind this one issued less than a full set
        MMD_BAR();
        __builtin_amdgcn_sched_barrier(0);
        if (s + LA < nsteps) issue(s + LA, std::integral_constant<int, nxt>{});
        __builtin_amdgcn_sched_barrier(0);
        const bf16_t* xs = lds + cur * XSLOT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ktl = wk * KS + ks;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (DBG == 1) {          // keep the W registers live without the matrix pipe
#pragma unroll
                    for (int j = 0; j < NT; ++j) if (i == 0) { const u32x4_t w4 = __builtin_bit_cast(u32x4_t, wb[cur][j][ks]); asm volatile("" :: "v"(w4[0]), "v"(w4[1]), "v"(w4[2]), "v"(w4[3])); }
                    continue;
                }
                const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(xs + (i * KSB + ktl) * 512 + aoff);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[cur][j][ks], xf, acc[i][j], 0, 0, 0);
            }
        }
    };
    int s = 0;
    for (; s + NB + LA <= nsteps; s += NB)          // steady periods: every step of the period and the LA - 1 steps behind each issue a full set
        stream_static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; step(s + u, std::integral_constant<int, u>{}, true); });
    for (; s < nsteps; )
        stream_static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; if (s < nsteps) { step(s, std::integral_constant<int, u>{}, s + LA < nsteps); ++s; } });

    // the WK partial sums of a column group meet in LDS (the ring is dead: every wave has read its last slot behind the barrier below)
    if constexpr (WK > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4_t*>(red + ((((wk - 1) * WN + wn) * MT + i) * NT + j) * 256 + lane * 4) = acc[i][j];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w2 = 0; w2 < WK - 1; ++w2)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += *reinterpret_cast<const f32x4_t*>(red + (((w2 * WN + wn) * MT + i) * NT + j) * 256 + lane * 4);
    }
    if (gridDim.y > 1 || p.slabs) {
        float* ws = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int m = i * 16 + lr, n = (nt0 + j) * 16 + lq * 4;
                if (m < p.M && nt0 + j < ntiles) {
                    f32x4_t v = acc[i][j];
                    if (p.slabs && p.wscale) v *= *reinterpret_cast<const f32x4_t*>(p.wscale + n);
                    *reinterpret_cast<f32x4_t*>(ws + (long long)m * p.N + n) = v;
                }
            }
        return;
    }
    // (the row loops below are compile-time recursions, not `#pragma unroll`: at MT = 16 the unrolled epilogue passes hipcc's pragma-unroll size limit, the loop stays rolled and `acc` moves to scratch for the whole kernel)
    if (p.epi == EPI_SWIGLU) {
        if constexpr (NT >= 2) {
            stream_static_for<MT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int m = i * 16 + lr;
#pragma unroll
                for (int j = 0; j < NT; j += 2) {
                    if (m < p.M && nt0 + j + 1 < ntiles) {
                        const int n_gate = (nt0 + j) * 16 + lq * 4;
                        *reinterpret_cast<s16x4_t*>((bf16_t*)p.Y + (long long)m * p.ldy + (n_gate >> 5) * 16 + (n_gate & 15)) = big_value_swiglu(acc[i][j], acc[i][j + 1], p.wscale, n_gate);
                    }
                }
            });
        }
        return;
    }
    stream_static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = i * 16 + lr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (nt0 + j < ntiles) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                store4<bf16_t>(p, m, (nt0 + j) * 16 + lq * 4, v);
            }
        }
    });
}

static inline void set_plan(const GemmArgs& a, int kernel, int tiles, int splits, int blocks) {
    if (a.plan_out) { a.plan_out[0] = kernel; a.plan_out[1] = tiles; a.plan_out[2] = splits; a.plan_out[3] = blocks; }
}

static void launch_gemv16(const GemmP& p, const GemmArgs& a, hipStream_t st) {
    const int KT = a.K >> 5, ntiles = a.N >> 4;
    int ksplit = 1;
    if (a.slabs_out && ntiles < 512 && KT >= 256 && a.splitk_ws) {          // long K, few n-tiles (down_proj): 2-4 K slabs
        ksplit = cdiv(768, ntiles); if (ksplit > 4) ksplit = 4;
        while (ksplit > 1 && (size_t)ksplit * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --ksplit;
    }
    else if (a.slabs_out && ntiles < 512 && a.splitk_ws) {
        // short K, few n-tiles (qkv, o at decode): one block per n-tile is latency-bound (8.0 / 6.4 us for 33 / 26 MB); two K slabs halve each wave's
        // dependent load chain: 6.7 / 5.3 us (fp8: 6.0 / 5.2 -> 4.7 / 4.1); the slab consumers sum them for free.  MMDUET_GEMV_KSPLIT_SHORT overrides (1..4)
        static const int ks_short = getenv("MMDUET_GEMV_KSPLIT_SHORT") ? atoi(getenv("MMDUET_GEMV_KSPLIT_SHORT")) : 2;
        ksplit = ks_short < 1 ? 1 : (ks_short > 4 ? 4 : ks_short);
        while (ksplit > 1 && (size_t)ksplit * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --ksplit;
    }
    if (a.slabs_out) *a.slabs_out = ksplit;
    set_plan(a, GEMM_K_GEMV16, ntiles, ksplit, (a.epi == EPI_SWIGLU || (ntiles % 2 == 0 && ntiles >= 2048) ? ntiles / 2 : ntiles) * ksplit);
    const bool two = a.epi == EPI_SWIGLU || (ntiles % 2 == 0 && ntiles >= 2048);
    const GemvChain none;
    const bool chain = a.chain && (a.chain->xn_h || a.chain->fin_h);
    const GemvChain& ch = chain ? *a.chain : none;
    GemmP q = p; if (a.Wp8) q.W = a.Wp8;
    if (chain && ch.fin_h) {
        // producer: one 16-wave block per n-tile, K split over the waves (no slabs, no second kernel)
        q.slabs = 0;
        if (a.slabs_out) *a.slabs_out = 0;
        set_plan(a, GEMM_K_GEMV16, ntiles, 1, ntiles);
        if (a.Wp8) hipLaunchKernelGGL((gemm_gemv16_kernel<1, 8, true, true, 16>), dim3(ntiles), dim3(1024), 0, st, q, KT, ch);
        else hipLaunchKernelGGL((gemm_gemv16_kernel<1, 8, false, true, 16>), dim3(ntiles), dim3(1024), 0, st, q, KT, ch);
        return;
    }
    const dim3 grid(two ? ntiles / 2 : ntiles, ksplit);
#define GEMV16_GO(NT_, W8_, CH_) hipLaunchKernelGGL((gemm_gemv16_kernel<NT_, 8, W8_, CH_>), grid, dim3(256), 0, st, q, KT, ch)
    if (a.Wp8) { if (two) { if (chain) GEMV16_GO(2, true, true); else GEMV16_GO(2, true, false); } else { if (chain) GEMV16_GO(1, true, true); else GEMV16_GO(1, true, false); } }
    else       { if (two) { if (chain) GEMV16_GO(2, false, true); else GEMV16_GO(2, false, false); } else { if (chain) GEMV16_GO(1, false, true); else GEMV16_GO(1, false, false); } }
#undef GEMV16_GO
}

// gemm_stream_kernel: 32 < M <= 256, packed bf16 weights, slab output (fused consumers) or the SwiGLU epilogue.  K must be a whole number of steps.
static bool stream_ok(int dtype, const GemmArgs& a) {
    if (dtype != MMD_BF16 || a.f16 || !a.Wp || a.M <= 32 || a.M > 256 || (a.N % 16) != 0 || (a.ldx % 8) != 0 || ((uintptr_t)a.X % 16) != 0 || a.out_f32) return false;
    if (a.Wp8 && a.M <= 64) return false;                                  // fp8 builds keep the 1-byte skinny kernel where it exists
    if ((long long)a.M * a.ldx * 2 >= (1ll << 32)) return false;            // X addressed as base + 32-bit offset
    const int ksb = a.M <= 128 ? 4 : 2;          // k-tiles per step of the instantiation that serves this M
    if ((a.K % (ksb * 32)) != 0) return false;
    if (a.epi == EPI_SWIGLU) return (a.N % 32) == 0 && !a.slabs_out;
    if (a.slabs_out) return a.splitk_ws != nullptr && a.epi == EPI_NONE && (size_t)a.M * a.N * sizeof(float) <= a.splitk_ws_bytes;          // (even ONE slab must fit)
    // epilogue in place (unfused schedule, several streams per forward): the same K split into the workspace, then the serial slab reduce applies bias / residual / activation --
    // slab for slab what the fused consumers do, so both schedules produce the same bits
    return a.splitk_ws != nullptr && (a.N % 4) == 0 && (a.ldy % 4) == 0 && (a.epi != EPI_RESID || (a.ldr % 4) == 0);
}
template <int MT, int NT, int WK, int KS, int NB, int DBG = 0, int WN = 4>
static hipError_t launch_stream_t(const GemmP& p, const GemmArgs& a, hipStream_t st, int want_split = 0) {
    constexpr int KSB = WK * KS;
    const int KT = a.K >> 5, ntiles = a.N >> 4;
    const int bx = cdiv(ntiles, WN * NT);
    int ksplit = 1;
    if (a.slabs_out || (a.epi != EPI_SWIGLU && a.splitk_ws)) {
        ksplit = want_split > 0 ? want_split : cdiv(288, bx);                                            // ~ one block per CU and a bit: every CU streams
        const int maxs = KT / (KSB * 6); if (ksplit > maxs) ksplit = maxs;  // >= 6 steps per block (the pipeline is NB - 1 steps deep)
        if (ksplit > 16) ksplit = 16; if (ksplit < 1) ksplit = 1;
        while (ksplit > 1 && (size_t)ksplit * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --ksplit;
    }
    const int ktper = (int)round_up(cdiv(KT, ksplit), KSB);
    ksplit = cdiv(KT, ktper);
    if (a.slabs_out) *a.slabs_out = ksplit;
    set_plan(a, GEMM_K_STREAM, ntiles, ksplit, bx * ksplit);
    constexpr size_t ring = (size_t)NB * MT * KSB * 1024, red = (size_t)(WK - 1) * WN * MT * NT * 1024;
    constexpr size_t smem = ring > red ? ring : red;
    static bool attr_set[64] = {};
    int adev = 0; hipGetDevice(&adev);
    if (smem > 65536 && adev >= 0 && adev < 64 && !attr_set[adev]) {
        hipFuncSetAttribute((const void*)gemm_stream_kernel<MT, NT, WK, KS, NB, DBG, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set[adev] = true;
    }
    hipLaunchKernelGGL((gemm_stream_kernel<MT, NT, WK, KS, NB, DBG, WN>), dim3(bx, ksplit), dim3(64 * WN * WK), smem, st, p, KT, ktper);
    if (!a.slabs_out && ksplit > 1) {          // epilogue in place: the serial slab reduce (slab 0, 1, 2, ... -- the fused consumers' order)
        long long work = (long long)a.M * ((a.N + 3) / 4);
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(cdiv(work, 256)), dim3(256), 0, st, p, ksplit);
    }
    return hipGetLastError();
}
// Decomposition of a streaming GEMM over the 256 CUs.  A CU streams ~25 GB/s whatever runs on it, so the launch is as long as its busiest CU: 296 four-wave blocks
// (gate_up at WN = 4) take two block rounds for 1.16 rounds of work -- 63 us where 237 five-pair blocks take 46 (tools/bench_gemm.py stream, profiles/r04_stream_sweep.txt).
// Pick the column-group width WN (n-tile slots per block = waves) and the K split that minimise   rounds x (W bytes + 0.5 X bytes per block) + slab bytes / 256
// under: whole K steps per block, >= 6 steps per block when K is split (the pipeline is NB - 1 steps deep), slabs within the workspace.
struct StreamPlan { int wn, ksplit; };
static StreamPlan stream_plan(const GemmArgs& a, int NT, int MT, int KSB, bool can_split) {
    const int KT = a.K >> 5, ntiles = a.N >> 4;
    StreamPlan best{4, 1}; double best_cost = 1e30;
    for (int wn = 4; wn <= 8; ++wn) {
        const int bx = cdiv(ntiles, wn * NT);
        for (int ks = 1; ks <= (can_split ? 16 : 1); ++ks) {
            const int ktper = (int)round_up(cdiv(KT, ks), KSB);
            if (cdiv(KT, ktper) != ks) continue;                                            // (this split count rounds to another one)
            if (ks > 1 && ktper < 6 * KSB) break;
            if (ks > 1 && (size_t)ks * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) break;
            const int rounds = cdiv((long long)bx * ks, 256);
            const double wb = (double)wn * NT * ktper * 1024, xb = (double)MT * 16 * ktper * 64;
            const double slab = can_split ? (double)ks * a.M * a.N * 8.0 / 256.0 : 0.0;
            const double cost = rounds * (wb + 0.5 * xb) + slab;
            if (cost < best_cost * 0.999) { best_cost = cost; best = StreamPlan{wn, ks}; }
        }
    }
    return best;
}
template <int MT, int NT, int WK, int KS, int NB>
static hipError_t launch_stream_wn(const GemmP& p, const GemmArgs& a, hipStream_t st) {
    const StreamPlan pl = stream_plan(a, NT, MT, WK * KS, a.slabs_out != nullptr || (a.epi != EPI_SWIGLU && a.splitk_ws != nullptr));
    switch (pl.wn) {
        case 5: return launch_stream_t<MT, NT, WK, KS, NB, 0, 5>(p, a, st, pl.ksplit);
        case 6: return launch_stream_t<MT, NT, WK, KS, NB, 0, 6>(p, a, st, pl.ksplit);
        case 7: return launch_stream_t<MT, NT, WK, KS, NB, 0, 7>(p, a, st, pl.ksplit);
        case 8: return launch_stream_t<MT, NT, WK, KS, NB, 0, 8>(p, a, st, pl.ksplit);
        default: return launch_stream_t<MT, NT, WK, KS, NB, 0, 4>(p, a, st, pl.ksplit);
    }
}
static hipError_t launch_stream(const GemmP& p, const GemmArgs& a, hipStream_t st) {
    const bool two = a.epi == EPI_SWIGLU || (a.N >> 4) >= 4096;
    if (a.M <= 64) return two ? launch_stream_wn<4, 2, 1, 4, 3>(p, a, st) : launch_stream_wn<4, 1, 1, 4, 3>(p, a, st);
    if (a.M <= 128) return two ? launch_stream_wn<8, 2, 1, 2, 4>(p, a, st) : launch_stream_wn<8, 1, 1, 4, 3>(p, a, st);
    return two ? launch_stream_wn<16, 2, 1, 1, 4>(p, a, st) : launch_stream_wn<16, 1, 1, 2, 4>(p, a, st);
}

template <int MT>
static void launch_skinny_mt(const GemmP& p, const GemmArgs& a, hipStream_t st) {
    const int KT = a.K >> 5, ntiles = a.N >> 4;
    const int NT = (a.epi == EPI_SWIGLU || ntiles >= 4096) ? 2 : 1;
    int nblocks = cdiv(ntiles, 4 * NT);
    int splits = 1;
    if (nblocks < 512 && a.epi != EPI_SWIGLU && a.splitk_ws) {
        splits = cdiv(512, nblocks);
        int maxs = KT / 16; if (maxs < 1) maxs = 1;              // >= 512 k per split
        if (splits > maxs) splits = maxs;
        if (splits > 8) splits = 8;
        while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --splits;
    }
    int ktper = (int)round_up(cdiv(KT, splits), 4);
    splits = cdiv(KT, ktper);
    dim3 grid(nblocks, splits);
    set_plan(a, GEMM_K_SKINNY, ntiles, splits, nblocks * splits);
    if (a.Wp8) {
        GemmP q = p; q.W = a.Wp8;
        if (NT == 2) hipLaunchKernelGGL((gemm_skinny_kernel<MT, 2, true>), grid, dim3(256), 0, st, q, KT, ktper);
        else hipLaunchKernelGGL((gemm_skinny_kernel<MT, 1, true>), grid, dim3(256), 0, st, q, KT, ktper);
    } else if (NT == 2) hipLaunchKernelGGL((gemm_skinny_kernel<MT, 2>), grid, dim3(256), 0, st, p, KT, ktper);
    else hipLaunchKernelGGL((gemm_skinny_kernel<MT, 1>), grid, dim3(256), 0, st, p, KT, ktper);
    if (a.slabs_out) { *a.slabs_out = splits; return; }
    if (splits > 1) {
        long long work = (long long)a.M * ((a.N + 3) / 4);
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(cdiv(work, 256)), dim3(256), 0, st, p, splits);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA-bound GEMM for M >= 128 rows (ViT tower / projector at M = 32 x 729, multi-frame LLM chunks at M = 49k).
// 128 x BN block tile, 4 waves (2 x 2), BK = 64, two LDS buffers filled by direct-to-LDS DMA (global_load_lds, 16 B per
// lane) one K-tile ahead of the MFMAs, ONE barrier per K-tile.
//   * W (packed, fragment-major): each 16x32 fragment tile is a contiguous 1 KiB piece -> one DMA instruction per piece,
//     the LDS image is already in fragment order (ds_read_b128 at lane*16: conflict-free, no swizzle).
//   * X (row-major): DMA pieces of 8 rows x 128 B (full cache lines); the 16-byte chunk index is XOR-ed with the row on
//     the SOURCE address and again on the fragment read (LDS destination of the DMA is lane-linear by construction),
//     which makes the row-strided fragment reads bank-conflict-free.
//   * rows beyond M are clamped on load and masked at the store; N % BN == 0 and K % 64 == 0 are dispatch conditions.
// ------------------------------------------------------------------------------------------------------------------
template <int BN, int EPI, bool F16 = false, int BM = 128>          // BM = 160 (round 5): 1274 rows x 56 column tiles = 448 blocks, two per CU in ONE round, instead of 560 (three rounds on 48 CUs)
__global__ __launch_bounds__(256) void gemm_big_kernel(GemmP p, int KT) {
    constexpr int BK = 64;
    static_assert(BM % 32 == 0, "two wave rows of whole 16-row tiles");
    constexpr int TM = BM / 32, TN = BN / 32;                // 16x16 tiles per wave (wave tile BM/2 x BN/2)
    constexpr int XE = BM * BK, WE = BN * BK;                // elements per buffer image
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (XE + WE)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous run of
    // n-tiles of one m-panel so the X panel stays in that XCD's L2
    const int nbx = gridDim.x, nby = gridDim.y;
    int bid = blockIdx.y * nbx + blockIdx.x;
    const int nblk = nbx * nby;
    {   // bijective for any grid size: XCD x owns a contiguous run of q (+1 for the first r XCDs) tile ids
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // tile order inside an XCD's run: sweep the dimension whose operand is SMALLER fastest, so the larger operand's panel
    // is fetched from HBM once and the smaller one is re-read from L2 / Infinity Cache.  ViT (M >> N): n fastest, the X
    // panel stays put; LLM chunks (N >> M, W = 270 MB > 256 MB Infinity Cache): m fastest, every W panel is streamed once
    // (measured FETCH_SIZE for gate_up at M = 980: 2.1 GB with n-fastest order = 8 x the weights).
    int mt, nt;
    if (p.N > p.M) { nt = bid / nby; mt = bid % nby; } else { mt = bid / nbx; nt = bid % nbx; }
    const int m0 = mt * BM, n0 = nt * BN;
    const bf16_t* X = (const bf16_t*)p.X;
    const bf16_t* Wp = (const bf16_t*)p.W;
    const int nsteps = p.K / BK;

    auto stage = [&](int buf, int step) {
        bf16_t* xs = lds + buf * (XE + WE);
        bf16_t* ws = xs + XE;
        const int k0 = step * BK;
#pragma unroll
        for (int j = 0; j < BM / 32; ++j) {                  // X: pieces of 8 rows x 64 k
            const int pi = wave + 4 * j;
            int row = m0 + pi * 8 + (lane >> 3);
            row = row < p.M ? row : p.M - 1;
            const int chunk = (lane & 7) ^ (lane >> 3);
            const bf16_t* src = X + (long long)row * p.ldx + k0 + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xs + pi * 512), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BN / 32; ++j) {                  // W: (n-tile, k-tile) fragment pieces
            const int wi = wave + 4 * j;
            const bf16_t* src = Wp + (((long long)(n0 / 16 + (wi >> 1)) * KT + (k0 >> 5) + (wi & 1)) * 64 + lane) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ws + wi * 512), 16, 0, 0);
        }
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0, 0, 0, 0};

    // optional split-K over grid.z (medium M with few tiles: fills the CUs; fp32 slabs reduced by splitk_reduce_kernel)
    const int zsteps = (nsteps + gridDim.z - 1) / gridDim.z;
    const int t0 = blockIdx.z * zsteps, t1 = min(nsteps, t0 + zsteps);
    if (t0 < t1) stage(0, t0);
    int cur = 0;
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                                     // (compiler drains the DMA queue here: vmcnt(0))
        if (t + 1 < t1) stage(cur ^ 1, t + 1);
        const bf16_t* xs = lds + cur * (XE + WE);
        const bf16_t* ws = xs + XE;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            bf16x8_t xf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mt = (wm >> 4) + i;
                const int c = kt * 4 + lq;
                xf[i] = *reinterpret_cast<const bf16x8_t*>(xs + (mt * 2 + (lr >> 3)) * 512 + (lr & 7) * 64 + ((c ^ (lr & 7)) * 8));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(ws + (((wn >> 4) + j) * 2 + kt) * 512 + lane * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<F16>(wf[j], xf[i], acc[i][j]);
        }
        cur ^= 1;
    }

    if (gridDim.z > 1) {
        float* ws = p.ws + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm + i * 16 + lr;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4_t*>(ws + (long long)m * p.N + n0 + wn + j * 16 + lq * 4) = acc[i][j];
            }
        }
        return;
    }
    // operands of the epilogue, fetched in ONE batch: per-column scale / bias quads (once per tile) and every residual quad of the wave's TM x TN tiles
    // (left to the compiler, each quad was loaded right before its use behind a full vmcnt(0): up to 24 serial round trips per tile)
    [[maybe_unused]] f32x4_t scq[TN]; [[maybe_unused]] s16x4_t biq[TN]; [[maybe_unused]] s16x4_t rq[TM][TN];
    const bool has_sc = p.wscale != nullptr, has_bi = p.bias != nullptr;
    if constexpr (EPI != EPI_SWIGLU) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + j * 16 + lq * 4;
            scq[j] = has_sc ? *reinterpret_cast<const f32x4_t*>(p.wscale + n) : f32x4_t{1, 1, 1, 1};
            biq[j] = has_bi ? *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.bias + n) : s16x4_t{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                rq[i][j] = s16x4_t{0, 0, 0, 0};
                if constexpr (EPI == EPI_RESID) rq[i][j] = *reinterpret_cast<const s16x4_t*>((const bf16_t*)p.R + (long long)min(m0 + wm + i * 16 + lr, p.M - 1) * p.ldr + n0 + wn + j * 16 + lq * 4);
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm + i * 16 + lr;
        [[maybe_unused]] const int mc = m < p.M ? m : p.M - 1;                      // rows past M compute on a valid row and are not stored
        if constexpr (EPI == EPI_SWIGLU) {
            if constexpr (TN == 4) {                                                // two output tiles: row-contiguous 16-byte stores (pair_to_row8)
                const int nb = n0 + wn, ob = (nb >> 5) * 16;
                const s16x8_t v = pair_to_row8(big_value_swiglu(acc[i][0], acc[i][1], p.wscale, nb + lq * 4), big_value_swiglu(acc[i][2], acc[i][3], p.wscale, nb + 32 + lq * 4));
                if (m < p.M) *reinterpret_cast<s16x8_t*>((bf16_t*)p.Y + (long long)m * p.ldy + ob + (lq & 1) * 16 + (lq >> 1) * 8) = v;
            } else {
#pragma unroll
                for (int j = 0; j < TN; j += 2) big_store_swiglu(p, m, n0 + wn + j * 16 + lq * 4, acc[i][j], acc[i][j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const int nb = n0 + wn + j * 16;
                const s16x8_t v = pair_to_row8(big_value_pre<EPI, F16>(acc[i][j], has_sc, scq[j], has_bi, biq[j], rq[i][j]), big_value_pre<EPI, F16>(acc[i][j + 1], has_sc, scq[j + 1], has_bi, biq[j + 1], rq[i][j + 1]));
                if (m < p.M) *reinterpret_cast<s16x8_t*>((bf16_t*)p.Y + (long long)m * p.ldy + nb + (lq & 1) * 16 + (lq >> 1) * 8) = v;
            }
        }
    }
}

// 160-row tiles of the 64-column form (plain / residual epilogue, bf16: a chunk's qkv / o_proj): when they take fewer block rounds per CU.  A CU holds three 128 x 64 blocks
// (48 KB of LDS each) or two 160 x 64 ones (56 KB); a CU's time ~ blocks it runs x rows per block.
static bool big_bm160(const GemmArgs& a, int BN) {
    if (BN != 64 || a.f16 || (a.epi != EPI_NONE && a.epi != EPI_RESID) || a.M < 512) return false;
    const long long t128 = (long long)(a.N / 64) * cdiv(a.M, 128), t160 = (long long)(a.N / 64) * cdiv(a.M, 160);
    if (t160 > 512) return false;                                          // (a third block per CU would have to wait for a slot)
    return (double)cdiv(t160, 256) * 160.0 < (double)cdiv(t128, 256) * 128.0 * 0.97;
}
template <int BN>
static void launch_big(const GemmP& p, const GemmArgs& a, hipStream_t st) {
    if (big_bm160(a, BN)) {
        if constexpr (BN == 64) {
            const int tiles = (a.N / 64) * cdiv(a.M, 160);
            set_plan(a, GEMM_K_BIG64, tiles, 1, tiles);
            const dim3 grid(a.N / 64, cdiv(a.M, 160), 1);
            if (a.epi == EPI_RESID) hipLaunchKernelGGL((gemm_big_kernel<64, EPI_RESID, false, 160>), grid, dim3(256), 0, st, p, a.K >> 5);
            else hipLaunchKernelGGL((gemm_big_kernel<64, EPI_NONE, false, 160>), grid, dim3(256), 0, st, p, a.K >> 5);
            return;
        }
    }
    const int tiles = (a.N / BN) * cdiv(a.M, 128);
    int splits = 1;
    if (tiles < 320 && a.K >= 8192 && a.epi != EPI_SWIGLU && a.splitk_ws && !a.f16) {      // long K, under one block per CU (down_proj): split K
        splits = 3;                                                                // (measured: splitting K = 3584 GEMMs costs more than it fills)
        if (tiles <= 64) splits = 4;
        while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --splits;
    } else if (tiles < 128 && a.K >= 2048 && a.epi != EPI_SWIGLU && a.splitk_ws && !a.f16) {
        // a handful of rows (65 <= M <= ~256: a few frames per forward, several streams' decode rows): 56-72 tiles cannot pull the weights out of HBM (a CU streams ~25 GB/s);
        // split K so that ~one block per CU streams.  round 3, 15 k context: an M = 98 step 8.5 -> see profiles/r03_decode_experiments.md
        splits = 256 / tiles; if (splits > 4) splits = 4; if (splits < 1) splits = 1;
        while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --splits;
    }
    dim3 grid(a.N / BN, cdiv(a.M, 128), splits);
    set_plan(a, BN == 128 ? GEMM_K_BIG128 : GEMM_K_BIG64, tiles, splits, tiles * splits);
    const int KT = a.K >> 5;
    if (a.f16) {          // the fp16 vision tower: plain / GELU(tanh) / residual epilogues
        switch (a.epi) {
            case EPI_GELU_TANH: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_GELU_TANH, true>), grid, dim3(256), 0, st, p, KT); break;
            case EPI_RESID: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_RESID, true>), grid, dim3(256), 0, st, p, KT); break;
            default: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_NONE, true>), grid, dim3(256), 0, st, p, KT); break;
        }
        return;
    }
    switch (a.epi) {
        case EPI_GELU_TANH: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_GELU_TANH>), grid, dim3(256), 0, st, p, KT); break;
        case EPI_GELU_ERF: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_GELU_ERF>), grid, dim3(256), 0, st, p, KT); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_RESID>), grid, dim3(256), 0, st, p, KT); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_SWIGLU>), grid, dim3(256), 0, st, p, KT); break;
        default: hipLaunchKernelGGL((gemm_big_kernel<BN, EPI_NONE>), grid, dim3(256), 0, st, p, KT); break;
    }
    if (splits > 1) {
        if (a.ring_slabs_out) { *a.ring_slabs_out = splits; return; }          // the caller folds the slabs itself (reduce + residual + RMSNorm in one pass)
        long long work = (long long)a.M * ((a.N + 3) / 4);
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(cdiv(work, 256)), dim3(256), 0, st, p, splits);
    }
}


// the ring kernels' masked output lanes need somewhere harmless to store (the store COUNT per wave must not depend on the tile): one 8 KB buffer per device
static void* ring_dump_slot() {
    static void* dump_slot[64] = {};
    int dev = 0; hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return nullptr;
    if (!dump_slot[dev] && hipMalloc(&dump_slot[dev], 8192) != hipSuccess) return nullptr;
    return dump_slot[dev];
}

template <int WN, bool M32, int NS, bool EARLY, bool F16 = false>
static hipError_t launch_ringx_t(const GemmP& p, const GemmArgs& a, hipStream_t st, int splits, bool stagger = true) {
    if (F16) splits = 1;
    while (splits > 1 && (a.epi == EPI_SWIGLU || !a.splitk_ws || (size_t)splits * a.M * a.N * sizeof(float) > a.splitk_ws_bytes)) --splits;
    constexpr int BN = 64 * WN;
    const int tiles = cdiv(a.N, BN) * cdiv(a.M, 256);
    const int slots = (WN == 2 && NS == 3) ? 512 : 256;                      // resident blocks: two 4-wave blocks per CU (72 KB rings), else one
    // ring_max_blocks: > 0 caps the persistent grid (tower share); < 0 (the overlap experiments of round 4, tools/probes/dropped/overlap_sweep.sh): NON-persistent, one tile per block, so that the
    // dispatcher can place another stream's blocks at every tile end
    const int cap = a.ring_max_blocks < 0 ? tiles : (a.ring_max_blocks > 0 && a.ring_max_blocks < slots ? a.ring_max_blocks : slots);
    dim3 grid(splits > 1 || tiles <= cap ? tiles : cap, 1, splits);
    set_plan(a, WN == 2 ? GEMM_K_RING128X2 : GEMM_K_RING256, tiles, splits, (int)grid.x * splits);
    const int KT = a.K >> 5;
    const size_t smem = NS * (256 * 32 + BN * 32) * sizeof(bf16_t);          // 96 KB / 72 KB (3 slots), 128 KB / 96 KB (4)
    static bool attr_set[64] = {};          // per device (hipFuncSetAttribute applies to the current device's code object)
    int adev = 0; hipGetDevice(&adev);
    if (adev >= 0 && adev < 64 && !attr_set[adev]) {
#define RX_ATTR(E) hipFuncSetAttribute((const void*)gemm_ringx_kernel<E, WN, M32, NS, EARLY, 0, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        RX_ATTR(EPI_NONE) RX_ATTR(EPI_GELU_TANH) RX_ATTR(EPI_RESID)
        if constexpr (!F16) { RX_ATTR(EPI_GELU_ERF) RX_ATTR(EPI_SWIGLU) }
#undef RX_ATTR
        attr_set[adev] = true;
    }
    const dim3 block(WN * 128);
    GemmP q = p;
    q.dump = ring_dump_slot();
    if (!q.dump) return hipErrorOutOfMemory;
    // start-up stagger of the second-slot blocks in ~4 us units: about half a tile (K/32 steps of ~0.75 us) -- see the kernel
    q.kper = (!stagger || splits > 1) ? 0 : ((a.K / 32) * 10) / 100 + 1;
    if constexpr (F16) {
        switch (a.epi) {
            case EPI_GELU_TANH: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_GELU_TANH, WN, M32, NS, EARLY, 0, true>), grid, block, smem, st, q, KT); break;
            case EPI_RESID: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_RESID, WN, M32, NS, EARLY, 0, true>), grid, block, smem, st, q, KT); break;
            case EPI_NONE: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_NONE, WN, M32, NS, EARLY, 0, true>), grid, block, smem, st, q, KT); break;
            default: return hipErrorInvalidValue;
        }
    } else
    switch (a.epi) {
        case EPI_GELU_TANH: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_GELU_TANH, WN, M32, NS, EARLY>), grid, block, smem, st, q, KT); break;
        case EPI_GELU_ERF: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_GELU_ERF, WN, M32, NS, EARLY>), grid, block, smem, st, q, KT); break;
        case EPI_RESID: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_RESID, WN, M32, NS, EARLY>), grid, block, smem, st, q, KT); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_SWIGLU, WN, M32, NS, EARLY>), grid, block, smem, st, q, KT); break;
        default: hipLaunchKernelGGL((gemm_ringx_kernel<EPI_NONE, WN, M32, NS, EARLY>), grid, block, smem, st, q, KT); break;
    }
    if (splits > 1) {
        if (a.ring_slabs_out) { *a.ring_slabs_out = splits; return hipGetLastError(); }
        long long work = (long long)a.M * ((a.N + 3) / 4);
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(cdiv(work, 256)), dim3(256), 0, st, p, splits);
    }
    return hipGetLastError();
}
#ifdef MMDUET_DEBUG_VARIANTS
template <int DBG>
static hipError_t launch_ringx_dbg(const GemmP& p, const GemmArgs& a, hipStream_t st) {
    const int tiles = cdiv(a.N, 256) * cdiv(a.M, 256);
    dim3 grid(tiles <= 256 ? tiles : 256, 1, 1);
    set_plan(a, GEMM_K_RING256, tiles, 1, (int)grid.x);
    const size_t smem = 3 * (256 * 32 + 256 * 32) * sizeof(bf16_t);
    hipFuncSetAttribute((const void*)gemm_ringx_kernel<EPI_NONE, 4, false, 3, true, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((gemm_ringx_kernel<EPI_NONE, 4, false, 3, true, DBG>), grid, dim3(512), smem, st, p, a.K >> 5);
    return hipGetLastError();
}
#endif
// flags: 16 = 8 waves, 256 x 256 tiles, three-slot ring, refill DMAs in the first rows of a step (every tower / projector GEMM, gate_up and split-K down of a chunk);
//        17 = 4 waves, 256 x 128 tiles, two blocks per CU (a chunk's qkv / o_proj where 256 x 256 tiles cannot fill the chip).  The other instantiations the template was
//        built to test (late refill, 32x32x16 MFMA, four slots) lost their A/B (header of gemm_ringx_kernel) and are no longer compiled into the library.
static hipError_t launch_ringx(int flags, const GemmP& p, const GemmArgs& a, hipStream_t st, int splits = 1) {
    if (a.f16) return launch_ringx_t<4, false, 3, true, true>(p, a, st, 1);          // the fp16 tower runs the shipped instantiation
    switch (flags & 27) {
        case 16: return launch_ringx_t<4, false, 3, true>(p, a, st, splits);
        case 17: return launch_ringx_t<2, false, 3, true>(p, a, st, splits);
        default: return hipErrorInvalidValue;
    }
}


// the ring GEMM addresses its operands as uniform base + 32-bit byte offset
static bool ring_size_ok(const GemmArgs& a) { return (long long)a.M * a.ldx * 2 < (1ll << 32) && (long long)a.N * a.K * 2 < (1ll << 32); }
// enough 256^2 tiles for the persistent ring: >= 400 (1.6 block waves of the 256 CUs), or close to whole waves from 0.75 of one up (4096^2: 256 tiles = one
// wave, 1.36 PF against 0.94 for the 128-row kernel; 300 tiles would leave the second wave at 17 % and stay with the 128-row kernel)
static bool ring_tiles_ok(long long t) { return t >= 400 || (t >= 192 && (double)t / (double)(cdiv((int)t, 256) * 256) >= 0.9); }
static bool big_packed_ok(int dtype, const GemmArgs& a, int BN) {
    return dtype == MMD_BF16 && a.Wp != nullptr && a.M > 64 && (a.N % BN) == 0 && (a.K % 64) == 0 && (a.ldx % 8) == 0 &&
           ((uintptr_t)a.X % 16) == 0 && !a.out_f32 && (a.ldy % 4) == 0 && ((uintptr_t)a.Y % 8) == 0 &&
           (a.epi != EPI_RESID || ((a.ldr % 4) == 0 && ((uintptr_t)a.R % 8) == 0)) && (a.bias == nullptr || ((uintptr_t)a.bias % 8) == 0);
}

// packed-W skinny path usable?  bf16, M <= 64, N % 16 == 0, K % 32 == 0, 16-byte aligned rows of X
static bool skinny_packed_ok(int dtype, const GemmArgs& a) {
    return dtype == MMD_BF16 && a.Wp != nullptr && a.M <= 64 && (a.N % 16) == 0 && (a.K % 32) == 0 && (a.ldx % 8) == 0 &&
           ((uintptr_t)a.X % 16) == 0 && (a.epi != EPI_SWIGLU || (a.N % 32) == 0);
}

// the two automatic ring conditions of launch_t (kept in one place: the model asks before it lays an activation out piece-major)
static bool ring256_auto(const GemmArgs& a) {
    return a.M >= 512 && (a.N % 32) == 0 && big_packed_ok(MMD_BF16, a, 16) && ring_size_ok(a) && ring_tiles_ok((long long)cdiv(a.M, 256) * cdiv(a.N, 256));
}
// K splits of the split-K ring for t256 output tiles.  Up to half a block wave of tiles: as many splits as fit one wave (down_proj of a chunk: 70 tiles x 3).  Between half a
// wave and the plain ring's threshold (down_proj of several streams' merged chunks: M = 2548 -> 140 tiles, which left 116 CUs idle for the whole K on the plain ring and ran at
// 0.28 of peak on the 128-row kernel) the split count comes from a small cost model: rounds of 256 blocks x K / sp steps of ~25 ns per unit of K, plus the fp32 slabs' write + read
// at ~5 TB/s (140 tiles: 3 splits = 420 items in two rounds of K / 3 -- two thirds of the unsplit time).
static int ring_split_choice(const GemmArgs& a) {
    const int t256 = cdiv(a.M, 256) * cdiv(a.N, 256);
    int sp = 256 / t256; if (sp < 1) sp = 1;
    if (t256 > 128) {
        double best = 1e30; int bsp = 1;
        for (int s = 1; s <= 8; ++s) {
            if (s > 1 && (a.K / s < 1024 || (size_t)s * a.M * a.N * sizeof(float) > a.splitk_ws_bytes)) break;
            const double rounds = (double)cdiv(t256 * s, 256);
            const double cost = rounds * ((double)a.K / s) * 0.025 + (s > 1 ? (double)s * a.M * a.N * 8.0 / 5e6 : 0.0);
            if (cost < best * 0.97) { best = cost; bsp = s; }          // (a finer split has to buy 3 %)
        }
        return bsp;
    }
    while (sp > 1 && a.K / sp < 1024) --sp;
    while (sp > 1 && (size_t)sp * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --sp;
    return sp;
}
static bool ring256_split_auto(const GemmArgs& a) {
    if (a.f16 || !(a.M >= 512 && a.K >= 8192 && big_packed_ok(MMD_BF16, a, 16) && (a.N % 32) == 0 && a.epi != EPI_SWIGLU && a.splitk_ws != nullptr && ring_size_ok(a))) return false;
    return ring_split_choice(a) >= 2;
}
bool gemm_ring_auto(int dtype, const GemmArgs& a, bool plain_only) {
    if (dtype != MMD_BF16 && dtype != MMD_F16) return false;
    GemmArgs b = a; b.f16 = dtype == MMD_F16;
    if (b.variant != GEMM_AUTO || b.wscale) return false;
    return ring256_auto(b) || (!plain_only && ring256_split_auto(b));
}
template <typename T>
static hipError_t launch_t(const GemmArgs& a, hipStream_t st, int* kind_out) {
    GemmP p;
    p.X = a.X; p.W = a.W; p.bias = a.bias; p.R = a.R; p.Y = a.Y; p.ws = a.splitk_ws; p.wscale = a.wscale;
    p.ldx = a.ldx; p.ldw = a.ldw; p.ldr = a.ldr; p.ldy = a.ldy;
    p.M = a.M; p.N = a.N; p.K = a.K; p.epi = a.epi; p.out_f32 = a.out_f32; p.slabs = a.slabs_out ? 1 : 0; p.dump = nullptr; p.flags = 0; p.kper = 0; p.x_pm = a.x_pm; p.y_pm = a.y_pm;
    p.vec = (sizeof(T) == 2 && (a.ldx % 8) == 0 && (a.ldw % 8) == 0 && ((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0) ? 1 : 0;
    if (a.ring_slabs_out) *a.ring_slabs_out = 0;
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    int variant = a.variant;
    bool skinny = (variant == GEMM_SKINNY) || (variant == GEMM_AUTO && a.M <= 64);
    bool large = (variant == GEMM_LARGE) || (variant == GEMM_AUTO && a.M >= 256 && a.N >= 128);
    if (kind_out) *kind_out = skinny ? MMD_K_GEMM_SKINNY : MMD_K_GEMM_TILE;
    if ((a.x_pm || a.y_pm) && (sizeof(T) != 2 || variant != GEMM_AUTO || !(ring256_auto(a) || ring256_split_auto(a)) || a.wscale ||
                               (a.x_pm && (a.K % 32)) || (a.y_pm && ((a.epi == EPI_SWIGLU ? a.N / 2 : a.N) % 32))))
        return hipErrorInvalidValue;          // a piece-major operand exists for the ring kernel only (the caller asks gemm_ring_auto first)
    if (a.y_pm && !ring256_auto(a)) return hipErrorInvalidValue;          // ... and a piece-major OUTPUT for its plain form only: the split-K form leaves fp32 slabs and splitk_reduce writes Y row-major
    if constexpr (sizeof(T) == 2) {
        // the weight-streaming regime above the GEMV's 16 rows: per-frame steps, short chunks (gemm_stream_kernel); slab consumers or the SwiGLU epilogue
        if ((variant == GEMM_AUTO || variant == GEMM_SKINNY || variant == GEMM_STREAM) && stream_ok(MMD_BF16, a)) {
            p.W = a.Wp;
            if (kind_out) *kind_out = MMD_K_GEMM_SKINNY;
            return launch_stream(p, a, st);
        }
        if (variant == GEMM_STREAM) return hipErrorInvalidValue;
#ifdef MMDUET_DEBUG_VARIANTS          // tools/bench_gemm.py stream: configuration sweep of gemm_stream_kernel at M <= 64 (`make DEBUG_VARIANTS=1`; never in the shipped library)
        if (variant >= 300 && variant < 340 && a.M <= 64 && a.Wp) {
            p.W = a.Wp; p.slabs = a.epi == EPI_SWIGLU ? 0 : 1;
            GemmArgs b = a; int dummy = 0; if (a.epi != EPI_SWIGLU) { b.slabs_out = &dummy; b.epi = EPI_NONE; p.epi = EPI_NONE; }
            const bool two = a.epi == EPI_SWIGLU;
#define SCFG(id, WK_, KS_, NB_, DBG_) case id: return two ? launch_stream_t<4, 2, WK_, KS_, NB_, DBG_>(p, b, st) : launch_stream_t<4, 1, WK_, KS_, NB_, DBG_>(p, b, st);
            switch (variant - 300) {
                SCFG(0, 2, 2, 4, 0) SCFG(1, 1, 4, 3, 0) SCFG(2, 4, 1, 4, 0) SCFG(3, 2, 4, 3, 0) SCFG(4, 1, 4, 4, 0) SCFG(5, 4, 2, 3, 0) SCFG(7, 2, 2, 6, 0)
                SCFG(10, 2, 2, 4, 1) SCFG(11, 1, 4, 3, 1) SCFG(12, 2, 2, 4, 2) SCFG(13, 1, 4, 3, 2)
                // balanced decompositions: 5 pairs per block (gate_up: 237 blocks), 7 n-tiles x 8 K chunks (down / o: 256 blocks)
                case 20: return two ? launch_stream_t<4, 2, 2, 2, 4, 0, 5>(p, b, st) : launch_stream_t<4, 1, 2, 2, 4, 0, 7>(p, b, st, 8);
                case 21: return two ? launch_stream_t<4, 2, 3, 2, 3, 0, 5>(p, b, st) : launch_stream_t<4, 1, 2, 2, 4, 0, 7>(p, b, st, 4);
                case 22: return two ? launch_stream_t<4, 2, 2, 2, 4, 1, 5>(p, b, st) : launch_stream_t<4, 1, 2, 2, 4, 1, 7>(p, b, st, 8);
                case 23: return two ? launch_stream_t<4, 2, 1, 4, 3, 0, 5>(p, b, st) : launch_stream_t<4, 1, 1, 4, 3, 0, 7>(p, b, st, 8);
                case 24: return two ? launch_stream_t<4, 2, 2, 2, 4, 0, 8>(p, b, st) : launch_stream_t<4, 1, 2, 2, 4, 0, 7>(p, b, st, 16);
                default: return hipErrorInvalidValue;
            }
#undef SCFG
        }
#endif
        // 256^2 tiles pay once there are ~1.5 block waves of them (every ViT / projector GEMM, gate_up of a >= 600-row chunk)
        if (variant == GEMM_RING256 || (variant == GEMM_AUTO && ring256_auto(a))) {
            if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0 || !ring_size_ok(a)) return hipErrorInvalidValue;
            p.W = a.Wp;
            if (kind_out) *kind_out = MMD_K_GEMM_TILE;
            return launch_ringx(a.ring_flags, p, a, st);
        }
        // long K with under one block wave of 256^2 tiles (down_proj of a chunk): split K across grid.z so ~one block per CU runs a
        // long steady state (1.05 PF at M = 1274 against 0.84 PF for the 128-row kernel's 3-way split); K = 3584 shapes lose to it
        const bool ring_split_ok = big_packed_ok(MMD_BF16, a, 16) && (a.N % 32) == 0 && a.epi != EPI_SWIGLU && a.splitk_ws != nullptr && ring_size_ok(a);
        if (!a.f16 && (variant == GEMM_RING256_SPLIT || (variant == GEMM_AUTO && a.M >= 512 && a.K >= 8192 && ring_split_ok))) {
            if (variant == GEMM_RING256_SPLIT && (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0 || !ring_size_ok(a))) return hipErrorInvalidValue;
            const int sp = ring_split_choice(a);
            if (variant == GEMM_RING256_SPLIT || sp >= 2) {
                p.W = a.Wp;
                if (kind_out) *kind_out = MMD_K_GEMM_TILE;
                return launch_ringx(16, p, a, st, sp);
            }
        }
#ifdef MMDUET_DEBUG_VARIANTS          // timing experiments of tools/bench_gemm.py (most give WRONG results): `make DEBUG_VARIANTS=1`; never in the shipped library
        if (variant == 92) {          // DBG 8 on the 4-slot ring
            if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0) return hipErrorInvalidValue; p.W = a.Wp;
            const int tiles = cdiv(a.N, 256) * cdiv(a.M, 256);
            const size_t smem = 4 * (256 * 32 + 256 * 32) * sizeof(bf16_t);
            hipFuncSetAttribute((const void*)gemm_ringx_kernel<EPI_NONE, 4, false, 4, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL((gemm_ringx_kernel<EPI_NONE, 4, false, 4, true, 8>), dim3(tiles <= 256 ? tiles : 256), dim3(512), smem, st, p, a.K >> 5);
            set_plan(a, GEMM_K_RING256, tiles, 1, tiles <= 256 ? tiles : 256);
            return hipGetLastError();
        }
        if (variant == 93) { if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0) return hipErrorInvalidValue; p.W = a.Wp; return launch_ringx_dbg<7>(p, a, st); }
        if (variant == 94) { if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0) return hipErrorInvalidValue; p.W = a.Wp; return launch_ringx_dbg<6>(p, a, st); }
        if (variant == 95) { if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0) return hipErrorInvalidValue; p.W = a.Wp; return launch_ringx_dbg<5>(p, a, st); }
        if (variant == 99) { if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0) return hipErrorInvalidValue; p.W = a.Wp; return launch_ringx_dbg<4>(p, a, st); }
        if (variant >= 96 && variant <= 98) {                               // timing experiments (WRONG results): see gemm_ringx_kernel DBG
            if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0) return hipErrorInvalidValue;
            p.W = a.Wp;
            return variant == 96 ? launch_ringx_dbg<1>(p, a, st) : (variant == 97 ? launch_ringx_dbg<2>(p, a, st) : launch_ringx_dbg<3>(p, a, st));
        }
#endif
        if (variant >= GEMM_RINGX && variant < GEMM_RINGX + 128) {          // forced ring variants (A/B and parity of every instantiation)
            if (!big_packed_ok(MMD_BF16, a, 16) || (a.N % 32) != 0 || !ring_size_ok(a)) return hipErrorInvalidValue;
            const int flags = variant - GEMM_RINGX;
            int sp = 1;
            if (flags & 4) {
                const int slots = (flags & 1) ? 512 : 256, tl = cdiv(a.M, 256) * cdiv(a.N, (flags & 1) ? 128 : 256);
                sp = slots / tl; if (sp < 1) sp = 1;
                while (sp > 1 && a.K / sp < 1024) --sp;
                if (sp < 2) sp = 2;
            }
            p.W = a.Wp;
            if (kind_out) *kind_out = MMD_K_GEMM_TILE;
            return launch_ringx(flags, p, a, st, sp);
        }
        const bool want_big = variant == GEMM_BIG || (variant == GEMM_AUTO && a.M > 64);
        if (want_big) {
            int bn = 0;
            // 128-wide tiles need ~1.5 block waves to keep two blocks per CU busy; below that 64-wide tiles (3 blocks/CU) win
            // by 8-10 % (measured at M = 980 / 1274, K = 3584); long-K shapes keep 128 and split K instead
            const long long t128 = (long long)cdiv(a.M, 128) * (a.N / 128);
            if (big_packed_ok(MMD_BF16, a, 128) && (t128 >= 400 || (a.K >= 8192 && t128 >= 224))) bn = 128;
            else if (big_packed_ok(MMD_BF16, a, 64)) bn = 64;
            // mid-M (a chunk's qkv / o_proj): once some CU would carry three or more 128-row blocks, one 256x128 ring tile per CU (4-wave ring, flags 17) is the
            // shorter schedule.  Per-CU cost in units of one 128x64 block at two per CU (17.5 us at K = 3584), fitted to tools/probes/midm_ring4w_sweep.py:
            // n blocks of 128x64 cost max(1.83, n), of 128x128 max(2.29, 1.77 n), a 256x128 ring tile 2.95 (M = 1323 qkv 75 -> 51 us, M = 1911 o 63 -> 58 us)
            if (bn && variant == GEMM_AUTO && !a.f16 && a.M >= 512 && a.K >= 1024 && (a.N % 128) == 0 && big_packed_ok(MMD_BF16, a, 16) && ring_size_ok(a)) {
                const long long mt128 = cdiv(a.M, 128);
                const double nb = bn == 64 ? (double)cdiv(mt128 * (a.N / 64), 256) : (double)cdiv(t128, 256);
                const double cbig = bn == 64 ? (nb > 1.83 ? nb : 1.83) : (1.77 * nb > 2.29 ? 1.77 * nb : 2.29);
                const double cr4 = 2.95 * (double)cdiv((long long)cdiv(a.M, 256) * (a.N / 128), 256);
                if (cr4 < 0.8 * cbig) {          // (only where the model predicts >= 20 %: inside the model, with each layer's weights cold, the 2-8 % cases of the sweep measured -0.3 %)
                    p.W = a.Wp;
                    if (kind_out) *kind_out = MMD_K_GEMM_TILE;
                    return launch_ringx(17, p, a, st);
                }
            }
            if (bn) {
                p.W = a.Wp;
                if (kind_out) *kind_out = MMD_K_GEMM_TILE;
                if (bn == 128) launch_big<128>(p, a, st); else launch_big<64>(p, a, st);
                return hipGetLastError();
            }
            if (variant == GEMM_BIG) return hipErrorInvalidValue;
        }
        if (a.f16) return hipErrorInvalidValue;          // IEEE-half operands exist in the ring / big kernels only (the tower's shapes: M >= 65, N % 64 == 0, K % 64 == 0, packed weights)
        if (skinny && skinny_packed_ok(MMD_BF16, a)) {
            p.W = a.Wp;
            if (a.M <= 16 && !a.no_gemv) launch_gemv16(p, a, st);
            else if (a.chain) return hipErrorInvalidValue;          // the decode chain exists in the GEMV kernel only
            else if (a.M <= 16) launch_skinny_mt<1>(p, a, st);
            else if (a.M <= 32) launch_skinny_mt<2>(p, a, st);
            else launch_skinny_mt<4>(p, a, st);
            return hipGetLastError();
        }
    }
    if (a.slabs_out) return hipErrorInvalidValue;          // slab mode exists only on the packed skinny path
    if (a.chain) return hipErrorInvalidValue;
    if (a.W == nullptr) return hipErrorInvalidValue;       // only the packed copy exists but the shape needs the generic path
    int splits = 1;
    if (skinny) {
        int blocks = cdiv(a.N, 64) * cdiv(a.M, 64);
        int want = cdiv(512, blocks);
        int maxs = a.K / 256; if (maxs < 1) maxs = 1;
        splits = want < maxs ? want : maxs;
        if (splits > 16) splits = 16;
        if (a.splitk_ws == nullptr) splits = 1;
        while (splits > 1 && (size_t)splits * a.M * a.N * sizeof(float) > a.splitk_ws_bytes) --splits;
    }
    int kper = (int)round_up(cdiv(a.K, splits), 32);
    splits = cdiv(a.K, kper);
    p.kper = kper;
    if (large) {
        dim3 grid(cdiv(a.N, 128), cdiv(a.M, 128), 1);
        set_plan(a, GEMM_K_TILE128, (int)(grid.x * grid.y), 1, (int)(grid.x * grid.y));
        hipLaunchKernelGGL((gemm_tile_kernel<T, 128, 128>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid(cdiv(a.N, 64), cdiv(a.M, 64), splits);
        set_plan(a, GEMM_K_TILE64, (int)(grid.x * grid.y), splits, (int)(grid.x * grid.y) * splits);
        hipLaunchKernelGGL((gemm_tile_kernel<T, 64, 64>), grid, dim3(256), 0, st, p);
        if (splits > 1) {
            long long work = (long long)a.M * ((a.N + 3) / 4);
            hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(cdiv(work, 256)), dim3(256), 0, st, p, splits);
        }
    }
    return hipGetLastError();
}

bool gemm_can_slab(int dtype, const GemmArgs& a) {
    if (skinny_packed_ok(dtype, a) && a.splitk_ws != nullptr && a.epi != EPI_SWIGLU) return true;
    int dummy = 0; GemmArgs b = a; b.slabs_out = &dummy; b.epi = EPI_NONE;
    return stream_ok(dtype, b);          // 64 < M <= 256: gemm_stream_kernel leaves slabs too
}

hipError_t launch_gemm(int dtype, const GemmArgs& a, hipStream_t st, int* kind_out) {
    if (dtype == MMD_F16) { GemmArgs h = a; h.f16 = 1; return launch_t<bf16_t>(h, st, kind_out); }          // 2-byte storage either way; the kernels' F16 forms read the bits as IEEE half
    return dtype == MMD_F32 ? launch_t<float>(a, st, kind_out) : launch_t<bf16_t>(a, st, kind_out);
}
