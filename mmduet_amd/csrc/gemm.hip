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

struct GemmP {
    const void* X; const void* W; const void* bias; const void* R; void* Y; float* ws;
    long long ldx, ldw, ldr, ldy;
    int M, N, K, epi, out_f32, kper, vec;
};

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
        if (n + r < p.N && p.bias) x += to_f<T>(((const T*)p.bias)[n + r]);
        if (p.epi == EPI_GELU_TANH) x = gelu_tanh_f(rnd<T>(x));
        else if (p.epi == EPI_GELU_ERF) x = gelu_erf_f(rnd<T>(x));
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
            float gg = rnd<T>(g[r]), uu = rnd<T>(u[r]);
            float s = rnd<T>(silu_f(gg));
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

template <typename T>
static hipError_t launch_t(const GemmArgs& a, hipStream_t st, int* kind_out) {
    GemmP p;
    p.X = a.X; p.W = a.W; p.bias = a.bias; p.R = a.R; p.Y = a.Y; p.ws = a.splitk_ws;
    p.ldx = a.ldx; p.ldw = a.ldw; p.ldr = a.ldr; p.ldy = a.ldy;
    p.M = a.M; p.N = a.N; p.K = a.K; p.epi = a.epi; p.out_f32 = a.out_f32;
    p.vec = (sizeof(T) == 2 && (a.ldx % 8) == 0 && (a.ldw % 8) == 0 && ((uintptr_t)a.X % 16) == 0 && ((uintptr_t)a.W % 16) == 0) ? 1 : 0;
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    int variant = a.variant;
    bool skinny = (variant == GEMM_SKINNY) || (variant == GEMM_AUTO && a.M <= 64);
    bool large = (variant == GEMM_LARGE) || (variant == GEMM_AUTO && a.M >= 256 && a.N >= 128);
    if (kind_out) *kind_out = skinny ? MMD_K_GEMM_SKINNY : MMD_K_GEMM_TILE;
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
        hipLaunchKernelGGL((gemm_tile_kernel<T, 128, 128>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid(cdiv(a.N, 64), cdiv(a.M, 64), splits);
        hipLaunchKernelGGL((gemm_tile_kernel<T, 64, 64>), grid, dim3(256), 0, st, p);
        if (splits > 1) {
            long long work = (long long)a.M * ((a.N + 3) / 4);
            hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(cdiv(work, 256)), dim3(256), 0, st, p, splits);
        }
    }
    return hipGetLastError();
}

hipError_t launch_gemm(int dtype, const GemmArgs& a, hipStream_t st, int* kind_out) {
    return dtype == MMD_F32 ? launch_t<float>(a, st, kind_out) : launch_t<bf16_t>(a, st, kind_out);
}
