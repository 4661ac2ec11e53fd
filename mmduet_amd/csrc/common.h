// common.h -- shared device helpers and the internal launcher interface of libmmduet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mmduet.h"

typedef uint16_t bf16_t;   // raw bfloat16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {            // round-to-nearest-even (v_cvt_pk_bf16_f32)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// ---- fp16 storage of the vision tower (config tower_dtype = fp16: the reference runs the tower under torch.cuda.amp.autocast(), models/modeling_live.py:28) ----
constexpr int MMD_F16 = 2;             // internal launcher dtype, never a context dtype: 2-byte IEEE half activations / weights, fp32 accumulate and statistics
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
__device__ __forceinline__ float h2f(uint16_t r) { return (float)__builtin_bit_cast(_Float16, r); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }          // round-to-nearest-even (v_cvt_f16_f32)
// raw 16-bit storage <-> float for kernels that move 2-byte elements as integer vectors
template <bool F16> __device__ __forceinline__ float raw2f(uint16_t r) { if constexpr (F16) return h2f(r); else return __uint_as_float(((uint32_t)r) << 16); }
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float to_f<f16_t>(f16_t v) { return (float)v; }
template <> __device__ __forceinline__ f16_t from_f<f16_t>(float v) { return (f16_t)v; }
template <bool F16> __device__ __forceinline__ uint16_t f2raw(float f) { if constexpr (F16) return f2h(f); else return f2bf(f); }
// one MFMA of the 2-byte GEMM / attention kernels: operands travel as raw 16-byte fragments, the instruction decides what the bits mean (same rate, same layouts)
template <bool F16> __device__ __forceinline__ f32x4_t mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// round a float through the storage type (the rounding points of eager bf16 execution)
template <typename T> __device__ __forceinline__ float rnd(float v) { return to_f<T>(from_f<T>(v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float k = 0.7978845608028654f;   // sqrt(2/pi)
    return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// fast forms for the bf16 path (error <= 2e-7 abs, far below bf16 resolution): raw v_exp_f32 / v_rcp_f32
__device__ __forceinline__ float gelu_tanh_fast(float x) {
    // 0.5 x (1 + tanh(u)) = x sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3):  2u log2(e) = x (A + B x^2).  Seven vector instructions (two transcendental) where the
    // 1 - 2 / (exp(2u) + 1) form took eleven -- the GELU epilogue is a fifth of fc1's tile time (K = 1152: 36 slices of MFMAs, then 128 elements per lane) --, and no
    // cancellation for very negative x; saturates correctly at +-inf (exp2 -> inf: rcp -> 0; exp2 -> 0: x).
    const float A = 2.3022081983f, B = 0.10294324f;            // 2 sqrt(2/pi) log2(e), A * 0.044715
    const float z = x * fmaf(x * x, B, A);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z));
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
    // erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
    float z = fabsf(x) * 0.7071067811865476f;
    float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    float er = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    er = x < 0.f ? -er : er;
    return 0.5f * x * (1.0f + er);
}
template <typename T> __device__ __forceinline__ float gelu_tanh_t(float x) { if constexpr (sizeof(T) == 2) return gelu_tanh_fast(x); else return gelu_tanh_f(x); }
template <typename T> __device__ __forceinline__ float gelu_erf_t(float x) { if constexpr (sizeof(T) == 2) return gelu_erf_fast(x); else return gelu_erf_f(x); }
template <typename T> __device__ __forceinline__ float silu_t(float x) {
    if constexpr (sizeof(T) == 2) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f)); else return silu_f(x);
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static inline size_t dtype_size(int dt) { return dt == MMD_F32 ? 4 : 2; }

// device-resident state of a graph-captured decode step (updated by the last kernel of the graph)
struct StepState { long long n_ctx; long long cap; void* K; void* V; int n_prev; int pad; };

// ---- epilogues of the GEMM family ------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_GELU_TANH = 1, EPI_GELU_ERF = 2, EPI_RESID = 3, EPI_SWIGLU = 4 };
enum { GEMM_AUTO = 0, GEMM_GENERIC = 1, GEMM_SKINNY = 2, GEMM_LARGE = 3, GEMM_BIG = 4, GEMM_SLAB = 5, GEMM_RING256 = 6, GEMM_RING256_SPLIT = 7,
       GEMM_RINGX = 16 /* + 1: 4-wave 256x128 blocks, + 2: 32x32x16 MFMA, + 4: split K */,
       GEMM_STREAM = 8 /* gemm_stream_kernel (32 < M <= 256, slabs or SwiGLU) */ };

// Decode chain of the weight-streaming GEMV (M <= 16, bf16): the residual add + RMSNorm between two GEMVs costs a launch and a cold, dependent
// load chain of its own (5.5 us + a kernel boundary, twice per layer at decode).  Instead
//   * the PRODUCER (o_proj / down_proj) owns an n-tile over all of K (one 16-wave block per tile, K split over the waves, LDS reduce) and folds
//     its result into the residual stream itself: h = rnd(rnd(x W^T) + h), leaving that tile's per-row sum of h^2 in ssq[m][tile];
//   * the CONSUMER (qkv / gate_up) never reads a normalised activation: every lane builds its X fragment as rnd(gamma * rnd(h * inv)),
//     inv = rsqrt(sum(ssq[m][:]) / K + eps) summed in a fixed order (deterministic; same rounding points as slab_resid_rmsnorm_kernel).
constexpr int GEMV_SSQ_STRIDE = 256;   // floats per row in ssq (n-tiles of 16 columns: N <= 4096)
constexpr int GEMV_CHAIN_ROWS = 4;     // rows a consumer keeps in LDS; K <= 4096
struct GemvChain {
    const void* xn_h = nullptr; const void* xn_gamma = nullptr; const float* xn_ssq = nullptr; float xn_eps = 0.f;     // consumer side
    void* fin_h = nullptr; float* fin_ssq = nullptr;                                                                     // producer side
};
struct GemmArgs {
    const void* X; int64_t ldx;      // [M,K]
    const void* W; int64_t ldw;      // [N,K]  (nn.Linear layout); may be null when only Wp exists
    const void* Wp = nullptr;        // same matrix, MFMA-fragment-major (launch_pack_w); enables the weight-streaming skinny kernel
    const void* Wp8 = nullptr;       // fp8 e4m3 copy of the (unscaled) weights, fragment-major in 64-k pairs (launch_pack_w8): the weight-streaming
                                     // kernels (M <= 64) read this one -- half the bytes; W / Wp then hold bf16(q), bit-identical values
    const float* wscale = nullptr;   // per-output-channel scale of a quantised matrix: Y = (X . q^T) * wscale[n] (+ bias ...)
    const void* bias;                // [N] or null (ctx dtype)
    const void* R; int64_t ldr;      // residual [M,N] (EPI_RESID)
    void* Y; int64_t ldy;            // [M,N] (or [M,N/2] for SWIGLU); ctx dtype, or fp32 if out_f32
    int M, N, K;
    int epi; int out_f32;
    int variant;
    float* splitk_ws; size_t splitk_ws_bytes;   // fp32 partial slabs
    int no_gemv = 0;                            // A/B switch: use the LDS-staged skinny kernel also for M <= 16
    int* ring_slabs_out = nullptr;              // if set: a split-K tile GEMM (ring / 128-row) leaves its [splits][M][N] fp32 slabs in splitk_ws and reports the count here
                                                // (0 = the GEMM ran unsplit and applied its epilogue itself); the caller consumes them with launch_slab_resid_rmsnorm
    int* slabs_out = nullptr;                   // if set (packed skinny path only): leave [splits][M][N] fp32 slabs in splitk_ws, no
                                                // epilogue, and return the split count here; a fused consumer kernel reduces them
    int ring_flags = 16;                        // ring GEMM instantiation the auto dispatch uses (launch_ringx flags; 16 = 8 waves, 256 x 256, early refill)
    int ring_max_blocks = 0;                    // > 0: cap on the persistent grid (co-residency experiments: leave CU resources to another stream)
    const GemvChain* chain = nullptr;           // gemv16 path only (M <= 16, packed bf16 / fp8 weights); see GemvChain
    int* plan_out = nullptr;                    // if set: int[4] = {kernel (GEMM_K_*), output tiles, K splits, blocks launched}
    int x_pm = 0, y_pm = 0;                     // X is / Y becomes a PIECE-MAJOR activation ([M/16][K/32] pieces of 16 rows x 32 elements, gemm_ringx_kernel): ring GEMMs only -- ask gemm_ring_auto first
    int f16 = 0;                                // operands, bias, residual and output are IEEE half (launch_gemm(MMD_F16, ...): the fp16 vision tower; ring / big kernels only)
};
// which kernel the dispatcher chose (mmd_op_gemm_last_plan; parity tests assert the production kernel really ran)
enum { GEMM_K_TILE64 = 0, GEMM_K_TILE128 = 1, GEMM_K_SKINNY = 2, GEMM_K_GEMV16 = 3, GEMM_K_BIG64 = 4, GEMM_K_BIG128 = 5, GEMM_K_RING256 = 6, GEMM_K_RING128X2 = 7, GEMM_K_STREAM = 8 };
bool gemm_can_slab(int dtype, const GemmArgs& a);
bool gemm_ring_auto(int dtype, const GemmArgs& a, bool plain_only = false);          // would the automatic dispatch run this GEMM on gemm_ringx_kernel (plain or split-K; plain_only: not the split-K form)?  (what a piece-major operand needs; a piece-major OUTPUT needs the plain form)

// launchers (dtype = mmd_dtype).  All return hipError_t of the launch.
hipError_t launch_gemm(int dtype, const GemmArgs& a, hipStream_t st, int* kind_out);
hipError_t launch_pack_w(const void* W, int64_t ldw, int N, int K, void* out, hipStream_t st);   // bf16 only, N%16==0, K%32==0
// fp8 e4m3 (OCP) weights, one scale per output channel: W [N,K] bf16 row-major is REPLACED by bf16(q) (q = rne_fp8(W / scale), scale = amax / 448),
// q8_rowmajor [N,K] bytes and scale [N] fp32 are written;  launch_pack_w8 lays q8 out fragment-major for the streaming kernels (N%16==0, K%64==0)
hipError_t launch_quantize_fp8_rows(void* W_bf16, int N, int K, uint8_t* q8_rowmajor, float* scale, hipStream_t st);
hipError_t launch_pack_w8(const uint8_t* q8_rowmajor, int N, int K, void* out, hipStream_t st);
hipError_t launch_rmsnorm(int dtype, const void* x, const void* w, void* y, int M, int H, float eps, hipStream_t st);
hipError_t launch_layernorm(int dtype, const void* x, const void* w, const void* b, void* y, int M, int H, float eps, hipStream_t st);
hipError_t launch_resid32_layernorm(const void* y16, float* h32, const void* pos16, int period, const void* ln_w, const void* ln_b, void* out16, void* outbf,
                                    int M, int H, float eps, hipStream_t st);          // the autocast tower's fp32 residual stream (ops.hip)
// fused consumers of skinny-GEMM slabs (ops.hip)
hipError_t launch_slab_resid_rmsnorm(const float* slabs, int splits, int M, int H, const void* resid_in, void* h_out, const void* norm_w, float eps,
                                     void* xn_out, hipStream_t st, const float* wscale = nullptr);
hipError_t launch_slab_rope_append(const float* slabs, int splits, const void* bias, int S, int nh, int nkv, int d, const float* inv_freq_dev,
                                   int64_t pos0, void* q_out, void* Kc, void* Vc, int64_t cap, hipStream_t st, const StepState* dyn = nullptr, int layer = 0,
                                   int slab_rows = 0);          // slab_rows: rows of one slab when the projection ran over more rows than this stream's S (0 = S); `slabs` points at the stream's first row
hipError_t launch_rope_table(void* tab, int S, int half, const float* inv_freq_dev, int64_t pos0, hipStream_t st, const StepState* dyn = nullptr);   // float2 [S][half], bf16-rounded
hipError_t launch_advance_state(StepState* st_dev, const int64_t* tok_dev, int64_t* prev_dev, int prev_cap, int64_t eos, int use_penalty, hipStream_t st);
hipError_t launch_add_rows(int dtype, void* x, const void* add, int M, int H, int period, hipStream_t st);   // x[m,:] += add[m % period,:]
hipError_t launch_rope_append_chunk(const void* qkv, int S, int nh, int nkv, const void* tab, int64_t pos0, void* q_out, void* Kc, void* Vc, int64_t cap, hipStream_t st);   // bf16, d = 128, transposed V, table from launch_rope_table
hipError_t launch_rope_append(int dtype, const void* qkv, int S, int nh, int nkv, int d, const float* inv_freq_dev, int64_t pos0, void* q_out,
                              void* Kc, void* Vc, int64_t cap, int v_transposed, hipStream_t st);
hipError_t launch_transpose_v(int dtype, const void* src, void* dst, int nkv, int64_t cap, int d, hipStream_t st);
hipError_t launch_embed(int dtype, const void* table, const int64_t* ids, int k, int H, int64_t vocab, void* out, hipStream_t st);
hipError_t launch_im2col(int dtype, const void* px, int B, int img, int patch, int grid, int Kpad, void* out, hipStream_t st);
hipError_t launch_gather_rows2(const void* xa, int64_t lda, void* ya, int Wa, const void* xb, int64_t ldb, void* yb, int Wb, const int32_t* rows_host, int n, hipStream_t st);   // n <= 64; rows ride in the kernel argument
hipError_t launch_pool(int dtype, const void* x, void* y, int B, int grid, int H, int mode, int stride, hipStream_t st);
hipError_t launch_gather_pool_rows(int dtype, const void* x, void* y, int B, int grid, int C, int out, hipStream_t st, int64_t ldx = 0);            // the (2 out)^2 tokens per frame a bilinear pool reads, compacted
hipError_t launch_pool_compact_bilinear(int dtype, const void* x, void* y, int B, int grid, int H, int out, hipStream_t st);      // the pool over that compact layout
hipError_t launch_heads(int dtype, const void* hidden, int64_t ldh, const int32_t* rows_dev, int M, const void* W4, int H, float* out,
                        hipStream_t st);
hipError_t launch_argmax_penalty(const float* logits, int V, const int64_t* prev_ids_dev, int n_prev, float penalty, int64_t* out_id,
                                 hipStream_t st, const StepState* dyn, void* scratch512);
// greedy sampling / token feed of several streams at once (mmd_round_multi): per-stream pointers ride in the kernel argument
constexpr int MMD_ROUND_MAX_SAMPLERS = 32;
struct SampleBatch { const int64_t* prev[MMD_ROUND_MAX_SAMPLERS]; int64_t* tok[MMD_ROUND_MAX_SAMPLERS]; int64_t* append[MMD_ROUND_MAX_SAMPLERS]; int n_prev[MMD_ROUND_MAX_SAMPLERS]; float penalty[MMD_ROUND_MAX_SAMPLERS]; };
struct FeedBatch { const int64_t* tok[MMD_ROUND_MAX_SAMPLERS]; int32_t row[MMD_ROUND_MAX_SAMPLERS]; };
hipError_t launch_sample_batch(const float* logits /*[n, V]*/, int V, const SampleBatch& b, int n, int64_t* toks_out_dev, void* scratch, hipStream_t st);
size_t sample_batch_scratch_bytes();
hipError_t launch_embed_feed(int dtype, const void* table, const FeedBatch& f, int n, int H, int64_t vocab, void* out, hipStream_t st);
hipError_t launch_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, hipStream_t st);
hipError_t launch_copy_rows(int dtype, const void* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols, hipStream_t st);
hipError_t launch_interleave16(int dtype, const void* a, const void* b, void* out, int rows, int cols, hipStream_t st);
hipError_t launch_pad_cols(int dtype, const void* src, int rows, int cols, void* dst, int cols_pad, hipStream_t st);
hipError_t launch_lora_merge(int dtype, void* W, const float* A, const float* B, int out_f, int in_f, int r, float scale, hipStream_t st);
hipError_t launch_preprocess_im2col(int dtype, const uint8_t* frames, int T, int R, int size, const int32_t* coef, const int32_t* bounds, int ksize,
                                    uint8_t* tmp, int patch, int grid, int Kpad, void* out, hipStream_t st);
hipError_t launch_normalize_frames(int dtype, const void* in, int in_kind /*0 uint8, 1 fp32*/, int B, int R, float rescale, const float* mean, const float* sd, void* out, hipStream_t st);
hipError_t launch_assemble_cls(int dtype, const void* patch, const void* cls, const void* pos, int B, int Tk, int C, void* out, hipStream_t st);
hipError_t launch_quick_gelu(int dtype, void* x, int64_t n, hipStream_t st);
hipError_t launch_letterbox(const uint8_t* src, int T, int H, int W, int new_w, int new_h, int R, int top, int left, const int32_t* xtab,
                            const int32_t* ytab, int area2x, int flip, uint32_t pad, uint8_t* dst, hipStream_t st);
hipError_t launch_preprocess(int dtype, const uint8_t* frames, int T, int R, int size, const int32_t* coef, const int32_t* bounds, int ksize,
                             uint8_t* tmp, void* out, hipStream_t st);

struct AttnArgs {
    const void* q; int64_t ldq;      // [S, nh*d] row stride ldq
    const void* K; const void* V;    // element (kv head h, token t, e) at K + h*k_hs + t*k_ts + e
    int64_t k_hs, k_ts, v_hs, v_ts;  // arena: hs = cap*d, ts = d ; fused ViT qkv rows: hs = d, ts = 3C
    int v_transposed;                // V arena stored transposed in 64-token blocks (see attn.hip); v_ts unused then
    void* out; int64_t ldo;          // [S, nh*d]
    int S, nh, nkv, d;
    int64_t n_ctx;                   // keys before this step; total keys = n_ctx + S
    int causal;
    int batch; int64_t q_bstride, kv_bstride, o_bstride;   // ViT: batch of independent sequences (elements)
    float* ws; size_t ws_bytes;      // split-KV partials
    int variant;
    const StepState* dyn; int layer; int dyn_splits;   // graph mode (attn_gqa128 only)
    // decode steps (rows <= 64, attn_gqa128<1> only): q / k / v are still the qkv projection's fp32 K-slabs.  The attention kernel itself reduces
    // them (+ bias), applies RoPE from the step's (cos, sin) table, builds its q fragments in registers and -- the block whose key range holds a new
    // position -- appends that token's K row / V column to the arena before staging the tile: no slab_rope_append launch, no q buffer.
    const float* qkv_slabs = nullptr; int n_slabs = 0; const void* qkv_bias = nullptr; const void* rope_tab = nullptr;   // slabs [n][slab_rows][(nh+2nkv)*d] (this step's first row); tab float2 [S][d/2]
    int slab_rows = 0;               // rows of one slab = the projection's M (0: S); larger than S when one GEMV served several streams' rows (mmd_round_multi)
    const StepState* segs = nullptr; int nseg = 0;   // launch_attention_decode_multi: the streams' (context length, capacity, arena base), device
};
hipError_t launch_attention(int dtype, const AttnArgs& a, hipStream_t st);
void attn_last_form(int* out2);          // {form, key splits} of the calling thread's most recent launch_attention* (attn.hip)
hipError_t launch_attention_decode_multi(const AttnArgs& a, hipStream_t st);     // decode rows of a.nseg streams in one launch (attn.hip)
