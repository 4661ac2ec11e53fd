// ops.hip -- the HBM-bound operators of the path: norms, RoPE + KV append, embedding gather, patch im2col, pooling,
// response heads, greedy sampling, layout helpers.  wave64 shuffles for reductions, 16-byte vector accesses.
#include "common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------------------------
// RMSNorm -- Qwen2RMSNorm.forward (transformers qwen2/modeling_qwen2.py:248-253): fp32 statistics,
// y = w * cast(x * rsqrt(mean(x^2) + eps)).  One wave per row.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int M, int H, float eps) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= M) return;
    const T* xr = x + (long long)row * H;
    T* yr = y + (long long)row * H;
    float ss = 0.f;
    if constexpr (sizeof(T) == 2) {
        if ((H & 7) == 0) {
            for (int c = lane * 8; c < H; c += 512) {
                s16x8_t v = *reinterpret_cast<const s16x8_t*>(xr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) { float f = bf2f((bf16_t)v[e]); ss += f * f; }
            }
        } else for (int c = lane; c < H; c += 64) { float f = to_f<T>(xr[c]); ss += f * f; }
    } else for (int c = lane; c < H; c += 64) { float f = to_f<T>(xr[c]); ss += f * f; }
    ss = wave_sum(ss);
    float inv = rsqrtf(ss / (float)H + eps);
    if constexpr (sizeof(T) == 2) {
        if ((H & 7) == 0) {
            for (int c = lane * 8; c < H; c += 512) {
                s16x8_t v = *reinterpret_cast<const s16x8_t*>(xr + c);
                s16x8_t g = *reinterpret_cast<const s16x8_t*>(w + c);
                s16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bf2f((bf16_t)g[e]) * bf2f(f2bf(bf2f((bf16_t)v[e]) * inv)));
                *reinterpret_cast<s16x8_t*>(yr + c) = o;
            }
            return;
        }
    }
    for (int c = lane; c < H; c += 64) yr[c] = from_f<T>(to_f<T>(w[c]) * rnd<T>(to_f<T>(xr[c]) * inv));
}

// the same for MANY rows (a chunk's ln2: 1274 x 3584): one 256-thread block per row, the row read ONCE into registers (16 bytes per thread and pass), 1274 blocks instead of
// 319 single-wave rows -- 7.7 -> ~4 us.  Same operations per element; the sum of squares is accumulated in another order (bf16-rounding-level differences, deterministic)
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int H, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = x + (long long)row * H;
    s16x8_t v[2]; float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (tid + 256 * j) * 8;
        v[j] = s16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (c < H) v[j] = *reinterpret_cast<const s16x8_t*>(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf2f((bf16_t)v[j][e]); ss += f * f; }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (tid + 256 * j) * 8;
        if (c < H) {
            const s16x8_t g = *reinterpret_cast<const s16x8_t*>(w + c);
            s16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(bf2f((bf16_t)g[e]) * bf2f(f2bf(bf2f((bf16_t)v[j][e]) * inv)));
            *reinterpret_cast<s16x8_t*>(y + (long long)row * H + c) = o;
        }
    }
}

hipError_t launch_rmsnorm(int dtype, const void* x, const void* w, void* y, int M, int H, float eps, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    if (dtype == MMD_BF16 && M >= 256 && (H & 7) == 0 && H <= 4096) {
        hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(M), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, H, eps);
        return hipGetLastError();
    }
    dim3 grid(cdiv(M, 4)), block(256);
    if (dtype == MMD_F32) hipLaunchKernelGGL(rmsnorm_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)w, (float*)y, M, H, eps);
    else hipLaunchKernelGGL(rmsnorm_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, M, H, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm -- nn.LayerNorm of SiglipEncoderLayer (siglip/modeling_siglip.py:329-331): fp32 mean / biased variance.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w, const T* __restrict__ b, T* __restrict__ y, int M, int H, float eps) {
    int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= M) return;
    const T* xr = x + (long long)row * H;
    T* yr = y + (long long)row * H;
    [[maybe_unused]] constexpr bool F16 = std::is_same<T, f16_t>::value;
    if constexpr (sizeof(T) == 2) {
        if ((H & 7) == 0 && H <= 2048) {
            // one wave per row, 16-byte accesses, the row stays in registers between the statistics and the output pass
            float v[4][8];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane * 8 + j * 512;
                if (c < H) {
                    s16x8_t raw = *reinterpret_cast<const s16x8_t*>(xr + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[j][e] = raw2f<F16>((uint16_t)raw[e]); s += v[j][e]; }
                }
            }
            const float mu = wave_sum(s) / (float)H;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane * 8 + j * 512;
                if (c < H) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dlt = v[j][e] - mu; q += dlt * dlt; }
                }
            }
            const float inv = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane * 8 + j * 512;
                if (c < H) {
                    s16x8_t g = *reinterpret_cast<const s16x8_t*>(w + c), bb = *reinterpret_cast<const s16x8_t*>(b + c), o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (short)f2raw<F16>((v[j][e] - mu) * inv * raw2f<F16>((uint16_t)g[e]) + raw2f<F16>((uint16_t)bb[e]));
                    *reinterpret_cast<s16x8_t*>(yr + c) = o;
                }
            }
            return;
        }
    }
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += to_f<T>(xr[c]);
    float mu = wave_sum(s) / (float)H;
    float v = 0.f;
    for (int c = lane; c < H; c += 64) { float d = to_f<T>(xr[c]) - mu; v += d * d; }
    float inv = rsqrtf(wave_sum(v) / (float)H + eps);
    for (int c = lane; c < H; c += 64) yr[c] = from_f<T>((to_f<T>(xr[c]) - mu) * inv * to_f<T>(w[c]) + to_f<T>(b[c]));
}

hipError_t launch_layernorm(int dtype, const void* x, const void* w, const void* b, void* y, int M, int H, float eps, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    dim3 grid(cdiv(M, 4)), block(256);
    if (dtype == MMD_F32) hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)w, (const float*)b, (float*)y, M, H, eps);
    else if (dtype == MMD_F16) hipLaunchKernelGGL(layernorm_kernel<f16_t>, grid, block, 0, st, (const f16_t*)x, (const f16_t*)w, (const f16_t*)b, (f16_t*)y, M, H, eps);
    else hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, M, H, eps);
    return hipGetLastError();
}

// The fp16 (autocast) tower's residual stream in fp32 -- models/modeling_live.py:28 runs the tower under torch.cuda.amp.autocast(): every linear returns
// fp16 (bias inside the op), LayerNorm is on the fp32 list, and `hidden + sublayer_out` promotes to fp32, so the hidden state between the fp16 matmuls
// is fp32.  One wave per row, the row in registers:
//     h32[row] = (pos ? float(pos[row % period]) : h32[row]) + float(y16[row])            (patch embeddings + position table, or residual + sublayer)
//     out16[row] = fp16(LayerNorm(h32[row]) * w + b)        when ln_w is given            (the next linear's input)
//     outbf[row] = bf16(h32[row])                           when outbf is given           (the tower's result: hidden_states[-1].to(images.dtype))
// y16 and out16 may be the same buffer (a row is read whole before it is written).
__global__ __launch_bounds__(256) void resid32_layernorm_kernel(const f16_t* y /* may alias out16: no __restrict__ on either */, float* __restrict__ h, const f16_t* __restrict__ pos, int period,
                                                                const f16_t* __restrict__ w, const f16_t* __restrict__ b, f16_t* out16, bf16_t* __restrict__ outbf,
                                                                int M, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const f16_t* yr = y + (long long)row * H;
    float* hr = h + (long long)row * H;
    const f16_t* pr = pos ? pos + (long long)(row % period) * H : nullptr;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < H) {
            const s16x8_t raw = *reinterpret_cast<const s16x8_t*>(yr + c);
            if (pr) {
                const s16x8_t pw = *reinterpret_cast<const s16x8_t*>(pr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] = h2f((uint16_t)pw[e]) + h2f((uint16_t)raw[e]);
            } else {
                const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(hr + c), a1 = *reinterpret_cast<const f32x4_t*>(hr + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[j][e] = a0[e] + h2f((uint16_t)raw[e]); v[j][e + 4] = a1[e] + h2f((uint16_t)raw[e + 4]); }
            }
            if (!outbf) {
                *reinterpret_cast<f32x4_t*>(hr + c) = f32x4_t{v[j][0], v[j][1], v[j][2], v[j][3]};
                *reinterpret_cast<f32x4_t*>(hr + c + 4) = f32x4_t{v[j][4], v[j][5], v[j][6], v[j][7]};
            } else {
                s16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(v[j][e]);
                *reinterpret_cast<s16x8_t*>(outbf + (long long)row * H + c) = o;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        }
    }
    if (!w) return;
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < H) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = v[j][e] - mu; q += dlt * dlt; }
        }
    }
    const float inv = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < H) {
            const s16x8_t g = *reinterpret_cast<const s16x8_t*>(w + c), bb = *reinterpret_cast<const s16x8_t*>(b + c);
            s16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (short)f2h((v[j][e] - mu) * inv * h2f((uint16_t)g[e]) + h2f((uint16_t)bb[e]));
            *reinterpret_cast<s16x8_t*>(out16 + (long long)row * H + c) = o;
        }
    }
}

hipError_t launch_resid32_layernorm(const void* y16, float* h32, const void* pos16, int period, const void* ln_w, const void* ln_b, void* out16, void* outbf,
                                    int M, int H, float eps, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    if ((H & 7) || H > 2048) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resid32_layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, (const f16_t*)y16, h32, (const f16_t*)pos16, period, (const f16_t*)ln_w, (const f16_t*)ln_b,
                       (f16_t*)out16, (bf16_t*)outbf, M, H, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Fused consumers of the skinny GEMM's split-K slabs (bf16 model path).  They replace, with identical rounding points,
//   splitk_reduce (+bias, +residual)  ->  RMSNorm          (after o_proj / down_proj)
//   splitk_reduce (+bias)             ->  RoPE + KV append (after the fused q/k/v projection)
// and save a launch + an activation round trip per GEMM.
// ---------------------------------------------------------------------------------------------------------------
template <int MAXS>
__global__ __launch_bounds__(256) void slab_resid_rmsnorm_kernel(const float* __restrict__ slabs, int splits, int M, int H, const bf16_t* __restrict__ resid,
                                                                 bf16_t* __restrict__ h_out, const bf16_t* __restrict__ w, float eps, bf16_t* __restrict__ xn,
                                                                 const float* __restrict__ wscale = nullptr) {          // wscale: per-column scale of an fp8-quantised matrix whose slabs are unscaled (tile GEMMs)
    // one block per row; thread t owns columns {4t + 1024 j}.  All slab loads of a column group are issued before the
    // first add (MAXS independent 16-byte loads in flight per thread) -- the kernel is pure load latency otherwise.
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const long long MN = (long long)M * H;
    float hv[4][4];                                    // up to H = 4096
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid * 4 + j * 1024;
        if (c < H) {
            const float* sp = slabs + (long long)m * H + c;
            f32x4_t part[MAXS];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) part[s] = s < splits ? *reinterpret_cast<const f32x4_t*>(sp + s * MN) : f32x4_t{0, 0, 0, 0};
            s16x4_t r = *reinterpret_cast<const s16x4_t*>(resid + (long long)m * H + c);
            f32x4_t acc = part[0];
#pragma unroll
            for (int s = 1; s < MAXS; ++s) acc += part[s];          // same order as the serial reduce: slab 0, 1, 2, ...
            if (wscale) acc *= *reinterpret_cast<const f32x4_t*>(wscale + c);          // (store4's order: scale, round, + residual)
            s16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = bf2f(f2bf(bf2f(f2bf(acc[e])) + bf2f((bf16_t)r[e])));      // rnd(rnd(gemm) + residual)
                hv[j][e] = v; ss += v * v; o[e] = (short)f2bf(v);
            }
            *reinterpret_cast<s16x4_t*>(h_out + (long long)m * H + c) = o;
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid * 4 + j * 1024;
        if (c < H) {
            s16x4_t g = *reinterpret_cast<const s16x4_t*>(w + c);
            s16x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (short)f2bf(bf2f((bf16_t)g[e]) * bf2f(f2bf(hv[j][e] * inv)));
            *reinterpret_cast<s16x4_t*>(xn + (long long)m * H + c) = o;
        }
    }
}
hipError_t launch_slab_resid_rmsnorm(const float* slabs, int splits, int M, int H, const void* resid_in, void* h_out, const void* norm_w, float eps,
                                     void* xn_out, hipStream_t st, const float* wscale) {
    if (M <= 0) return hipSuccess;
    if (H > 4096 || (H & 3) || splits > 16) return hipErrorInvalidValue;
    if (splits <= 4) hipLaunchKernelGGL(slab_resid_rmsnorm_kernel<4>, dim3(M), dim3(256), 0, st, slabs, splits, M, H, (const bf16_t*)resid_in, (bf16_t*)h_out, (const bf16_t*)norm_w, eps, (bf16_t*)xn_out, wscale);
    else if (splits <= 8) hipLaunchKernelGGL(slab_resid_rmsnorm_kernel<8>, dim3(M), dim3(256), 0, st, slabs, splits, M, H, (const bf16_t*)resid_in, (bf16_t*)h_out, (const bf16_t*)norm_w, eps, (bf16_t*)xn_out, wscale);
    else hipLaunchKernelGGL(slab_resid_rmsnorm_kernel<16>, dim3(M), dim3(256), 0, st, slabs, splits, M, H, (const bf16_t*)resid_in, (bf16_t*)h_out, (const bf16_t*)norm_w, eps, (bf16_t*)xn_out, wscale);
    return hipGetLastError();
}

__global__ void slab_rope_append_kernel(const float* __restrict__ slabs, int splits, const bf16_t* __restrict__ bias, int S, int nh, int nkv, int d,
                                        const float* __restrict__ inv_freq_tab, long long pos0, bf16_t* __restrict__ q_out, bf16_t* __restrict__ Kc,
                                        bf16_t* __restrict__ Vc, long long cap, const StepState* __restrict__ dyn, int layer, int slab_rows) {
    if (dyn) {                                     // graph replay: position / arena come from device state
        pos0 = dyn->n_ctx; cap = dyn->cap;
        const long long le = (long long)nkv * cap * d;
        Kc = (bf16_t*)dyn->K + layer * le; Vc = (bf16_t*)dyn->V + layer * le;
    }
    const int s = blockIdx.x, head = blockIdx.y, half = d >> 1;
    const int row_w = (nh + 2 * nkv) * d;
    const long long MN = (long long)slab_rows * row_w;
    const long long base = (long long)s * row_w + (long long)head * d;
    const long long pos = pos0 + s;
    auto val = [&](int i) {                            // rnd(sum of slabs + bias): the bf16 output of the q/k/v projection
        float part[16];
#pragma unroll
        for (int z = 0; z < 16; ++z) part[z] = z < splits ? slabs[z * MN + base + i] : 0.f;      // independent loads first
        float a = part[0];
#pragma unroll
        for (int z = 1; z < 16; ++z) a += part[z];
        return bf2f(f2bf(a + bf2f(bias[head * d + i])));
    };
    if (head >= nh + nkv) {
        const int kvh = head - nh - nkv;
        bf16_t* dst = Vc + (long long)kvh * cap * d + ((pos >> 6) * d << 6) + (pos & 63);      // transposed 64-token blocks
        for (int i = threadIdx.x; i < d; i += blockDim.x) dst[(long long)i << 6] = f2bf(val(i));
        return;
    }
    bf16_t* dst = head < nh ? q_out + (long long)s * nh * d + (long long)head * d : Kc + ((long long)(head - nh) * cap + pos) * d;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float ang = (float)pos * inv_freq_tab[i];
        float c = bf2f(f2bf(cosf(ang))), sn = bf2f(f2bf(sinf(ang)));
        float x1 = val(i), x2 = val(i + half);
        float o1 = bf2f(f2bf(x1 * c)) + bf2f(f2bf(-x2 * sn));
        float o2 = bf2f(f2bf(x2 * c)) + bf2f(f2bf(x1 * sn));
        dst[i] = f2bf(o1);
        dst[i + half] = f2bf(o2);
    }
}
// (cos, sin) of the step's positions, bf16-rounded as the RoPE consumers use them: one table per step, shared by all layers and heads
// (the decode attention kernel's fused q / k preparation reads it instead of evaluating cosf / sinf per layer)
__global__ void rope_table_kernel(float2* __restrict__ tab, int half, const float* __restrict__ inv_freq_tab, long long pos0, const StepState* __restrict__ dyn) {
    if (dyn) pos0 = dyn->n_ctx;
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float ang = (float)(pos0 + s) * inv_freq_tab[i];
        tab[s * half + i] = float2{bf2f(f2bf(cosf(ang))), bf2f(f2bf(sinf(ang)))};
    }
}
hipError_t launch_rope_table(void* tab, int S, int half, const float* inv_freq_dev, int64_t pos0, hipStream_t st, const StepState* dyn) {
    if (S <= 0) return hipSuccess;
    hipLaunchKernelGGL(rope_table_kernel, dim3(S), dim3(64), 0, st, (float2*)tab, half, inv_freq_dev, (long long)pos0, dyn);
    return hipGetLastError();
}
hipError_t launch_slab_rope_append(const float* slabs, int splits, const void* bias, int S, int nh, int nkv, int d, const float* inv_freq_dev,
                                   int64_t pos0, void* q_out, void* Kc, void* Vc, int64_t cap, hipStream_t st, const StepState* dyn, int layer, int slab_rows) {
    if (S <= 0) return hipSuccess;
    hipLaunchKernelGGL(slab_rope_append_kernel, dim3(S, nh + 2 * nkv), dim3(64), 0, st, slabs, splits, (const bf16_t*)bias, S, nh, nkv, d, inv_freq_dev,
                       (long long)pos0, (bf16_t*)q_out, (bf16_t*)Kc, (bf16_t*)Vc, (long long)cap, dyn, layer, slab_rows > 0 ? slab_rows : S);
    return hipGetLastError();
}

// x[m, :] += add[m % period, :]   (learned position embedding, siglip/modeling_siglip.py:184)
template <typename T>
__global__ void add_rows_kernel(T* __restrict__ x, const T* __restrict__ add, long long total, int H, int period) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long m = i / H; int c = (int)(i % H);
    x[i] = from_f<T>(to_f<T>(x[i]) + to_f<T>(add[(m % period) * H + c]));
}
hipError_t launch_add_rows(int dtype, void* x, const void* add, int M, int H, int period, hipStream_t st) {
    long long total = (long long)M * H;
    if (total <= 0) return hipSuccess;
    dim3 grid(cdiv(total, 256)), block(256);
    if (dtype == MMD_F32) hipLaunchKernelGGL(add_rows_kernel<float>, grid, block, 0, st, (float*)x, (const float*)add, total, H, period);
    else if (dtype == MMD_F16) hipLaunchKernelGGL(add_rows_kernel<f16_t>, grid, block, 0, st, (f16_t*)x, (const f16_t*)add, total, H, period);
    else hipLaunchKernelGGL(add_rows_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)x, (const bf16_t*)add, total, H, period);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// RoPE + KV append -- Qwen2RotaryEmbedding + apply_rotary_pos_emb + cache update (qwen2/modeling_qwen2.py:84-135,
// 218-222).  cos/sin are computed in fp32 from position = pos0 + s, then rounded to the storage type; the rotate-half
// form q*cos + rotate_half(q)*sin is evaluated with the reference's rounding (each product, then the sum).
// Input: fused qkv rows [S][(nh + 2 nkv) d].  Output: q_out [S][nh d]; K/V arena [nkv][cap][d] at token pos0 + s.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_append_kernel(const T* __restrict__ qkv, int S, int nh, int nkv, int d, const float* __restrict__ inv_freq_tab, long long pos0,
                                   T* __restrict__ q_out, T* __restrict__ Kc, T* __restrict__ Vc, long long cap, int v_tr) {
    int s = blockIdx.x;
    int head = blockIdx.y;                 // 0..nh-1 q heads, nh..nh+nkv-1 k heads, then v heads
    int half = d >> 1;
    int row_w = (nh + 2 * nkv) * d;
    const T* src = qkv + (long long)s * row_w + (long long)head * d;
    long long pos = pos0 + s;
    if (head >= nh + nkv) {               // V: plain copy into the arena
        int kvh = head - nh - nkv;
        if (v_tr) {     // transposed 64-token blocks: (tok, e) at ((tok>>6)*d + e)*64 + (tok&63)
            T* dst = Vc + (long long)kvh * cap * d + ((pos >> 6) * d << 6) + (pos & 63);
            for (int i = threadIdx.x; i < d; i += blockDim.x) dst[(long long)i << 6] = src[i];
        } else {
            T* dst = Vc + ((long long)kvh * cap + pos) * d;
            for (int i = threadIdx.x; i < d; i += blockDim.x) dst[i] = src[i];
        }
        return;
    }
    T* dst = head < nh ? q_out + (long long)s * nh * d + (long long)head * d : Kc + ((long long)(head - nh) * cap + pos) * d;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float ang = (float)pos * inv_freq_tab[i];       // fp32 product, like the reference's fp32 outer product
        float c = rnd<T>(cosf(ang)), sn = rnd<T>(sinf(ang));
        float x1 = to_f<T>(src[i]), x2 = to_f<T>(src[i + half]);
        // out[i] = x1*cos + (-x2)*sin ; out[i+half] = x2*cos + x1*sin
        float o1 = rnd<T>(x1 * c) + rnd<T>(-x2 * sn);
        float o2 = rnd<T>(x2 * c) + rnd<T>(x1 * sn);
        dst[i] = from_f<T>(o1);
        dst[i + half] = from_f<T>(o2);
    }
}

// The same for a CHUNK (bf16, head_dim 128, transposed V arena; S = hundreds of tokens): 16-byte loads / stores, (cos, sin) from the step's table
// (launch_rope_table: evaluated once per step instead of per layer and head), V through an LDS transpose so that the arena's 64-token rows are written as
// rows.  Same arithmetic and rounding points as rope_append_kernel (the table holds rnd(cosf), rnd(sinf) of the same fp32 angle).  One launch:
// blocks [0, S) rotate q and k of one token (256 threads = 32 heads x 8 lanes; more heads loop), blocks [S, S + nblk * nkv) transpose one
// (64-token arena block, kv head) of v.  The scalar kernel above took 16 us per layer at S = 1274 (cosf / sinf per element and head, 2-byte scatter stores).
__global__ __launch_bounds__(256) void rope_append_chunk_kernel(const bf16_t* __restrict__ qkv, int S, int nh, int nkv, const float2* __restrict__ tab, long long pos0,
                                                                bf16_t* __restrict__ q_out, bf16_t* __restrict__ Kc, bf16_t* __restrict__ Vc, long long cap) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) bf16_t vt[64 * (D + 8)];
    const int tid = threadIdx.x;
    const int row_w = (nh + 2 * nkv) * D;
    if ((int)blockIdx.x < S) {
        const int s = blockIdx.x, part = tid & 7;
        const long long pos = pos0 + s;
        f32x4_t cs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cs[u] = *reinterpret_cast<const f32x4_t*>(tab + (long long)s * (D / 2) + part * 8 + u * 2);      // (cos, sin) of elements part*8 + 2u, + 2u + 1
        for (int head = tid >> 3; head < nh + nkv; head += 32) {
            const bf16_t* src = qkv + (long long)s * row_w + (long long)head * D + part * 8;
            const s16x8_t a = *reinterpret_cast<const s16x8_t*>(src), b = *reinterpret_cast<const s16x8_t*>(src + D / 2);
            s16x8_t o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = cs[e >> 1][(e & 1) * 2], sn = cs[e >> 1][(e & 1) * 2 + 1];
                const float x1 = bf2f((bf16_t)a[e]), x2 = bf2f((bf16_t)b[e]);
                o1[e] = (short)f2bf(bf2f(f2bf(x1 * c)) + bf2f(f2bf(-x2 * sn)));
                o2[e] = (short)f2bf(bf2f(f2bf(x2 * c)) + bf2f(f2bf(x1 * sn)));
            }
            bf16_t* dst = head < nh ? q_out + (long long)s * nh * D + (long long)head * D : Kc + ((long long)(head - nh) * cap + pos) * D;
            *reinterpret_cast<s16x8_t*>(dst + part * 8) = o1;
            *reinterpret_cast<s16x8_t*>(dst + D / 2 + part * 8) = o2;
        }
        return;
    }
    // v: arena block b (positions [64 b, 64 b + 64)) of kv head kvh; tokens of this step inside it: [lo, hi)
    const int vb = blockIdx.x - S;
    const int kvh = vb % nkv;
    const long long b = (pos0 >> 6) + vb / nkv;
    const long long lo = b * 64 > pos0 ? b * 64 : pos0, hi = (b + 1) * 64 < pos0 + S ? (b + 1) * 64 : pos0 + S;
    for (int i = tid; i < 64 * (D / 8); i += 256) {              // [token][16-byte chunk]
        const int t = i >> 4, ch = i & 15;
        const long long pos = b * 64 + t;
        if (pos >= lo && pos < hi)
            *reinterpret_cast<s16x8_t*>(vt + t * (D + 8) + ch * 8) = *reinterpret_cast<const s16x8_t*>(qkv + (pos - pos0) * row_w + (long long)(nh + nkv + kvh) * D + ch * 8);
    }
    __syncthreads();
    bf16_t* dstb = Vc + (long long)kvh * cap * D + ((b * D) << 6);          // (tok, e) at ((tok >> 6) * d + e) * 64 + (tok & 63)
    for (int i = tid; i < D * 8; i += 256) {                     // [e][group of 8 tokens]
        const int e = i >> 3, g8 = i & 7;
        const long long p0 = b * 64 + g8 * 8;
        if (p0 >= lo && p0 + 8 <= hi) {
            s16x8_t o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (short)vt[(g8 * 8 + k) * (D + 8) + e];
            *reinterpret_cast<s16x8_t*>(dstb + ((long long)e << 6) + g8 * 8) = o;
        } else {
            for (int k = 0; k < 8; ++k) if (p0 + k >= lo && p0 + k < hi) dstb[((long long)e << 6) + g8 * 8 + k] = vt[(g8 * 8 + k) * (D + 8) + e];
        }
    }
}
hipError_t launch_rope_append_chunk(const void* qkv, int S, int nh, int nkv, const void* tab, int64_t pos0, void* q_out, void* Kc, void* Vc, int64_t cap, hipStream_t st) {
    if (S <= 0) return hipSuccess;
    const int nblk = (int)(((pos0 + S - 1) >> 6) - (pos0 >> 6) + 1);
    hipLaunchKernelGGL(rope_append_chunk_kernel, dim3(S + nblk * nkv), dim3(256), 0, st, (const bf16_t*)qkv, S, nh, nkv, (const float2*)tab, (long long)pos0,
                       (bf16_t*)q_out, (bf16_t*)Kc, (bf16_t*)Vc, (long long)cap);
    return hipGetLastError();
}

hipError_t launch_rope_append(int dtype, const void* qkv, int S, int nh, int nkv, int d, const float* l2, int64_t pos0, void* q_out,
                              void* Kc, void* Vc, int64_t cap, int v_tr, hipStream_t st) {
    if (S <= 0) return hipSuccess;
    dim3 grid(S, nh + 2 * nkv), block(64);
    if (dtype == MMD_F32) hipLaunchKernelGGL(rope_append_kernel<float>, grid, block, 0, st, (const float*)qkv, S, nh, nkv, d, l2, (long long)pos0, (float*)q_out, (float*)Kc, (float*)Vc, (long long)cap, v_tr);
    else hipLaunchKernelGGL(rope_append_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)qkv, S, nh, nkv, d, l2, (long long)pos0, (bf16_t*)q_out, (bf16_t*)Kc, (bf16_t*)Vc, (long long)cap, v_tr);
    return hipGetLastError();
}

// V arena layout conversion (parity-test entry points): [nkv][cap][d] row-major -> transposed 64-token blocks
template <typename T>
__global__ void transpose_v_kernel(const T* __restrict__ src, T* __restrict__ dst, long long cap, int d, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int e = (int)(i % d); long long rest = i / d; long long tok = rest % cap; long long h = rest / cap;
    dst[h * cap * d + ((tok >> 6) * d + e) * 64 + (tok & 63)] = src[i];
}
hipError_t launch_transpose_v(int dtype, const void* src, void* dst, int nkv, int64_t cap, int d, hipStream_t st) {
    long long total = (long long)nkv * cap * d;
    if (total <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(transpose_v_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)src, (float*)dst, (long long)cap, d, total);
    else hipLaunchKernelGGL(transpose_v_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, (long long)cap, d, total);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// embedding gather -- nn.Embedding (qwen2/modeling_qwen2.py:322); ids clamped like joint_embed (modeling_live.py:44)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(const T* __restrict__ table, const int64_t* __restrict__ ids, int H, long long vocab, T* __restrict__ out) {
    long long id = ids[blockIdx.x];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const T* src = table + id * H;
    T* dst = out + (long long)blockIdx.x * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) dst[c] = src[c];
}
hipError_t launch_embed(int dtype, const void* table, const int64_t* ids, int k, int H, int64_t vocab, void* out, hipStream_t st) {
    if (k <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(embed_kernel<float>, dim3(k), dim3(256), 0, st, (const float*)table, ids, H, (long long)vocab, (float*)out);
    else hipLaunchKernelGGL(embed_kernel<bf16_t>, dim3(k), dim3(256), 0, st, (const bf16_t*)table, ids, H, (long long)vocab, (bf16_t*)out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// patch im2col -- SiglipVisionEmbeddings conv(k = stride = patch) as a GEMM operand (siglip/modeling_siglip.py:124-179):
// row (b, gy, gx), column (c, ky, kx) in conv-weight flatten order, zero-padded to Kpad.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, typename TO = T>          // TO != T: bf16 pixel_values into the fp16 tower's patch matrix (exact: |x| <= 1 and 8 mantissa bits)
__global__ void im2col_kernel(const T* __restrict__ px, int img, int patch, int grid, int Kpad, TO* __restrict__ out) {
    int row = blockIdx.x;                              // b*grid*grid + gy*grid + gx
    int b = row / (grid * grid), rem = row % (grid * grid), gy = rem / grid, gx = rem % grid;
    int K = 3 * patch * patch;
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
        TO v = 0;
        if (k < K) {
            int c = k / (patch * patch), r2 = k % (patch * patch), ky = r2 / patch, kx = r2 % patch;
            const T s = px[(((long long)b * 3 + c) * img + gy * patch + ky) * img + gx * patch + kx];
            if constexpr (std::is_same<T, TO>::value) v = s; else v = from_f<TO>(to_f<T>(s));
        }
        out[(long long)row * Kpad + k] = v;
    }
}
hipError_t launch_im2col(int dtype, const void* px, int B, int img, int patch, int grid, int Kpad, void* out, hipStream_t st) {
    int rows = B * grid * grid;
    if (rows <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(rows), dim3(256), 0, st, (const float*)px, img, patch, grid, Kpad, (float*)out);
    else if (dtype == MMD_F16) hipLaunchKernelGGL(im2col_kernel<f16_t>, dim3(rows), dim3(256), 0, st, (const f16_t*)px, img, patch, grid, Kpad, (f16_t*)out);
    else if (dtype == MMD_F16 + 1) hipLaunchKernelGGL((im2col_kernel<bf16_t, f16_t>), dim3(rows), dim3(256), 0, st, (const bf16_t*)px, img, patch, grid, Kpad, (f16_t*)out);
    else hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, (const bf16_t*)px, img, patch, grid, Kpad, (bf16_t*)out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// post_projector_pooling (models/live_llava/video_head_live_llava_qwen.py:100-119) on token-major [B, g*g, H]:
// bilinear = F.interpolate(size=ceil(g/stride), align_corners=False); average/max = pool2d(stride).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilinear_tap(int o, int n_in, int n_out, int& i0, int& i1, float& lam) {
    float scale = (float)n_in / (float)n_out;
    float src = ((float)o + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)floorf(src);
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    lam = src - (float)i0;
}
template <typename T>
__global__ void pool_kernel(const T* __restrict__ x, T* __restrict__ y, int g, int H, int mode, int stride, int out) {
    int b = blockIdx.z, oy = blockIdx.y / out, ox = blockIdx.y % out;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    const T* xb = x + (long long)b * g * g * H;
    float r;
    if (mode == MMD_POOL_BILINEAR) {
        int y0, y1, x0, x1; float ly, lx;
        bilinear_tap(oy, g, out, y0, y1, ly);
        bilinear_tap(ox, g, out, x0, x1, lx);
        float a = to_f<T>(xb[((long long)y0 * g + x0) * H + c]), bq = to_f<T>(xb[((long long)y0 * g + x1) * H + c]);
        float cq = to_f<T>(xb[((long long)y1 * g + x0) * H + c]), dq = to_f<T>(xb[((long long)y1 * g + x1) * H + c]);
        float top = a * (1.f - lx) + bq * lx, bot = cq * (1.f - lx) + dq * lx;
        r = top * (1.f - ly) + bot * ly;
    } else if (mode == MMD_POOL_ADAPTIVE_AVG) {
        // adaptive_avg_pool2d of the token grid (models/vision_live.py:17-24): bin i = [floor(i g / out), ceil((i+1) g / out))
        const int ys = (oy * g) / out, ye = ((oy + 1) * g + out - 1) / out, xs = (ox * g) / out, xe = ((ox + 1) * g + out - 1) / out;
        float acc = 0.f;
        for (int yy = ys; yy < ye; ++yy)
            for (int xx = xs; xx < xe; ++xx) acc += to_f<T>(xb[((long long)yy * g + xx) * H + c]);
        r = acc / (float)((ye - ys) * (xe - xs));
    } else {
        float acc = mode == MMD_POOL_MAX ? -INFINITY : 0.f;
        for (int dy = 0; dy < stride; ++dy)
            for (int dx = 0; dx < stride; ++dx) {
                float v = to_f<T>(xb[((long long)(oy * stride + dy) * g + ox * stride + dx) * H + c]);
                acc = mode == MMD_POOL_MAX ? fmaxf(acc, v) : acc + v;
            }
        r = mode == MMD_POOL_MAX ? acc : acc / (float)(stride * stride);
    }
    y[(((long long)b * out + oy) * out + ox) * H + c] = from_f<T>(r);
}
// The bilinear pool (F.interpolate, no antialias) reads only the 2 x 2 neighbours of each of its out^2 sample points: (2 out)^2 = 196 of a frame's 729 tokens at the
// shipped sizes.  The projector in front of it is row-wise, so it only has to run on those tokens -- the pooled result is the same function of the same values (the
// reference computes the other 533 rows of `connector` and never reads them; same reasoning as the lazy lm_head).  Compact row u = (2 oy + ty) * 2 out + (2 ox + tx)
// holds source token (tap_ty(oy), tap_tx(ox)); a tap pair that clamps at the border simply appears twice.
template <typename T>
__global__ void gather_pool_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int g, int C, int out, long long ldx) {      // ldx: source row stride (C, or 3 C for the q third of a fused qkv row)
    const int b = blockIdx.y, u = blockIdx.x, two = 2 * out;
    const int ry = u / two, rx = u % two;
    int y0, y1, x0, x1; float l;
    bilinear_tap(ry >> 1, g, out, y0, y1, l); bilinear_tap(rx >> 1, g, out, x0, x1, l);
    const int tok = ((ry & 1) ? y1 : y0) * g + ((rx & 1) ? x1 : x0);
    const T* src = x + ((long long)b * g * g + tok) * ldx;
    T* dst = y + ((long long)b * two * two + u) * C;
    if ((C * sizeof(T)) % 16 == 0) { for (int i = threadIdx.x; i < (int)(C * sizeof(T) / 16); i += blockDim.x) reinterpret_cast<u32x4_t*>(dst)[i] = reinterpret_cast<const u32x4_t*>(src)[i]; }
    else for (int i = threadIdx.x; i < C; i += blockDim.x) dst[i] = src[i];
}
template <typename T>
__global__ void pool_compact_bilinear_kernel(const T* __restrict__ x, T* __restrict__ y, int g, int H, int out) {
    const int b = blockIdx.z, oy = blockIdx.y / out, ox = blockIdx.y % out, two = 2 * out;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    const T* xb = x + (long long)b * two * two * H;
    int y0, y1, x0, x1; float ly, lx;
    bilinear_tap(oy, g, out, y0, y1, ly); bilinear_tap(ox, g, out, x0, x1, lx);          // (only the weights are needed here; the rows sit at fixed compact positions)
    const float a = to_f<T>(xb[((long long)(2 * oy) * two + 2 * ox) * H + c]), bq = to_f<T>(xb[((long long)(2 * oy) * two + 2 * ox + 1) * H + c]);
    const float cq = to_f<T>(xb[((long long)(2 * oy + 1) * two + 2 * ox) * H + c]), dq = to_f<T>(xb[((long long)(2 * oy + 1) * two + 2 * ox + 1) * H + c]);
    const float top = a * (1.f - lx) + bq * lx, bot = cq * (1.f - lx) + dq * lx;
    y[(((long long)b * out + oy) * out + ox) * H + c] = from_f<T>(top * (1.f - ly) + bot * ly);
}
hipError_t launch_gather_pool_rows(int dtype, const void* x, void* y, int B, int grid, int C, int out, hipStream_t st, int64_t ldx) {
    if (B <= 0 || out <= 0) return hipSuccess;
    if (ldx <= 0) ldx = C;
    dim3 g(4 * out * out, B), block(64);
    if (dtype == MMD_F32) hipLaunchKernelGGL(gather_pool_rows_kernel<float>, g, block, 0, st, (const float*)x, (float*)y, grid, C, out, (long long)ldx);
    else hipLaunchKernelGGL(gather_pool_rows_kernel<bf16_t>, g, block, 0, st, (const bf16_t*)x, (bf16_t*)y, grid, C, out, (long long)ldx);          // (any 2-byte type: bf16 / f16 rows are copied as bits)
    return hipGetLastError();
}
hipError_t launch_pool_compact_bilinear(int dtype, const void* x, void* y, int B, int grid, int H, int out, hipStream_t st) {
    if (B <= 0 || out <= 0) return hipSuccess;
    dim3 g(cdiv(H, 256), out * out, B), block(256);
    if (dtype == MMD_F32) hipLaunchKernelGGL(pool_compact_bilinear_kernel<float>, g, block, 0, st, (const float*)x, (float*)y, grid, H, out);
    else hipLaunchKernelGGL(pool_compact_bilinear_kernel<bf16_t>, g, block, 0, st, (const bf16_t*)x, (bf16_t*)y, grid, H, out);
    return hipGetLastError();
}

// ya[i, :] = xa[rows[i], :], yb[i, :] = xb[rows[i], :]  (2-byte elements; the rows of a chunk's last decoder layer that anything downstream reads).  The row list
// travels as a kernel argument: no staging copy, nothing for a later step to overwrite.  grid = (n, 2).
struct NeedRows { int32_t r[64]; };
__global__ void gather_rows2_kernel(const unsigned short* __restrict__ xa, long long lda, unsigned short* __restrict__ ya, int Wa,
                                    const unsigned short* __restrict__ xb, long long ldb, unsigned short* __restrict__ yb, int Wb, NeedRows rows) {
    const bool second = blockIdx.y != 0;
    const long long ldx = second ? ldb : lda; const int W = second ? Wb : Wa;
    const unsigned short* src = (second ? xb : xa) + (long long)rows.r[blockIdx.x] * ldx;
    unsigned short* dst = (second ? yb : ya) + (long long)blockIdx.x * W;
    if ((W & 7) == 0 && (ldx & 7) == 0) { for (int i = threadIdx.x; i < W / 8; i += blockDim.x) reinterpret_cast<u32x4_t*>(dst)[i] = reinterpret_cast<const u32x4_t*>(src)[i]; }
    else for (int i = threadIdx.x; i < W; i += blockDim.x) dst[i] = src[i];
}
hipError_t launch_gather_rows2(const void* xa, int64_t lda, void* ya, int Wa, const void* xb, int64_t ldb, void* yb, int Wb, const int32_t* rows_host, int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 64) return hipErrorInvalidValue;
    NeedRows rows; for (int i = 0; i < 64; ++i) rows.r[i] = rows_host[i < n ? i : n - 1];
    hipLaunchKernelGGL(gather_rows2_kernel, dim3(n, 2), dim3(256), 0, st, (const unsigned short*)xa, (long long)lda, (unsigned short*)ya, Wa,
                       (const unsigned short*)xb, (long long)ldb, (unsigned short*)yb, Wb, rows);
    return hipGetLastError();
}

hipError_t launch_pool(int dtype, const void* x, void* y, int B, int grid, int H, int mode, int stride, hipStream_t st) {
    int out = mode == MMD_POOL_BILINEAR ? (grid + stride - 1) / stride : mode == MMD_POOL_ADAPTIVE_AVG ? stride : grid / stride;   // adaptive: `stride` carries the output side
    if (B <= 0 || out <= 0) return hipSuccess;
    dim3 g(cdiv(H, 256), out * out, B), block(256);
    if (dtype == MMD_F32) hipLaunchKernelGGL(pool_kernel<float>, g, block, 0, st, (const float*)x, (float*)y, grid, H, mode, stride, out);
    else hipLaunchKernelGGL(pool_kernel<bf16_t>, g, block, 0, st, (const bf16_t*)x, (bf16_t*)y, grid, H, mode, stride, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// informative_head / relevance_head (models/live_llava/video_head_live_llava_qwen.py:77-78,160-161): [4,H] x selected rows,
// output rounded through the storage type (the reference's GEMM output dtype) then widened: `.float()`.
// One wave per (row, head output).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void heads_kernel(const T* __restrict__ hidden, long long ldh, const int32_t* __restrict__ rows, const T* __restrict__ W4, int H, float* __restrict__ out) {
    int m = blockIdx.x, o = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int row = rows ? rows[m] : m;
    const T* h = hidden + (long long)row * ldh;
    const T* w = W4 + (long long)o * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += to_f<T>(h[c]) * to_f<T>(w[c]);
    s = wave_sum(s);
    if (lane == 0) out[m * 4 + o] = rnd<T>(s);
}
hipError_t launch_heads(int dtype, const void* hidden, int64_t ldh, const int32_t* rows_dev, int M, const void* W4, int H, float* out, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(heads_kernel<float>, dim3(M), dim3(256), 0, st, (const float*)hidden, (long long)ldh, rows_dev, (const float*)W4, H, out);
    else hipLaunchKernelGGL(heads_kernel<bf16_t>, dim3(M), dim3(256), 0, st, (const bf16_t*)hidden, (long long)ldh, rows_dev, (const bf16_t*)W4, H, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// greedy sampling: RepetitionPenaltyLogitsProcessor (score/p if score > 0 else score*p on seen ids, transformers
// generation/logits_process.py) + argmax (first maximal index, like torch.argmax) -- models/modeling_live.py:60-72.
// Single block; V up to a few 100k.
// ---------------------------------------------------------------------------------------------------------------
// two stages (V = 152064: one block would crawl through 600 KB alone): ARGMAX_BLOCKS blocks reduce their slice to a
// (value, index) candidate, one block picks the winner; ties -> smallest index, like torch.argmax
#define ARGMAX_BLOCKS 64
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void argmax_penalty_kernel(const float* __restrict__ logits, int V, const int64_t* __restrict__ prev, int n_prev, float penalty,
                                                             float* __restrict__ cand_v, int* __restrict__ cand_i, const StepState* __restrict__ dyn) {
    __shared__ float sv[4]; __shared__ int si[4];
    if (dyn && penalty != 1.f) n_prev = dyn->n_prev;
    const int per = (V + gridDim.x - 1) / gridDim.x;
    const int beg = blockIdx.x * per, end = min(V, beg + per);
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
        float v = logits[i];
        if (n_prev > 0) {
            bool seen = false;
            for (int j = 0; j < n_prev; ++j) if (prev[j] == i) { seen = true; break; }
            if (seen) v = v < 0.f ? v * penalty : v / penalty;
        }
        if (better(v, i, best, bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (better(sv[k], si[k], best, bi)) { best = sv[k]; bi = si[k]; }
        cand_v[blockIdx.x] = best; cand_i[blockIdx.x] = bi;
    }
}
__global__ void argmax_final_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i, int n, int64_t* __restrict__ out_id) {
    float best = -INFINITY; int bi = 0x7fffffff;
    const int lane = threadIdx.x;
    if (lane < n) { best = cand_v[lane]; bi = cand_i[lane]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) *out_id = bi;
}
hipError_t launch_argmax_penalty(const float* logits, int V, const int64_t* prev_ids_dev, int n_prev, float penalty, int64_t* out_id, hipStream_t st,
                                 const StepState* dyn, void* scratch) {
    float* cand_v = (float*)scratch; int* cand_i = (int*)((char*)scratch + ARGMAX_BLOCKS * sizeof(float));      // >= 512 bytes of context scratch
    hipLaunchKernelGGL(argmax_penalty_kernel, dim3(ARGMAX_BLOCKS), dim3(256), 0, st, logits, V, prev_ids_dev, n_prev, penalty, cand_v, cand_i, dyn);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(64), 0, st, cand_v, cand_i, ARGMAX_BLOCKS, out_id);
    return hipGetLastError();
}

// The same for several streams in ONE pass (mmd_round_multi): row r of logits [n, V] against sampler r's own penalty list.  The winner goes to the sampler's token slot
// (the next round's embedding gather reads it there), to toks_out[r] (one host copy for all streams) and -- penalty on -- behind the sampler's list: the host counts that
// entry only when the token was not EOS (models/modeling_live.py:66-72), a stale entry beyond the count is never read.
__global__ __launch_bounds__(256) void argmax_penalty_multi_kernel(const float* __restrict__ logits, int V, SampleBatch b, float* __restrict__ cand_v, int* __restrict__ cand_i) {
    __shared__ float sv[4]; __shared__ int si[4];
    const int r = blockIdx.y;
    const float* lg = logits + (long long)r * V;
    const int64_t* prev = b.prev[r]; const int n_prev = b.n_prev[r]; const float penalty = b.penalty[r];
    const int per = (V + gridDim.x - 1) / gridDim.x;
    const int beg = blockIdx.x * per, end = min(V, beg + per);
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
        float v = lg[i];
        if (n_prev > 0) {
            bool seen = false;
            for (int j = 0; j < n_prev; ++j) if (prev[j] == i) { seen = true; break; }
            if (seen) v = v < 0.f ? v * penalty : v / penalty;
        }
        if (better(v, i, best, bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (better(sv[k], si[k], best, bi)) { best = sv[k]; bi = si[k]; }
        cand_v[r * ARGMAX_BLOCKS + blockIdx.x] = best; cand_i[r * ARGMAX_BLOCKS + blockIdx.x] = bi;
    }
}
__global__ void argmax_final_multi_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i, SampleBatch b, int64_t* __restrict__ toks_out) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float best = -INFINITY; int bi = 0x7fffffff;
    if (lane < ARGMAX_BLOCKS) { best = cand_v[r * ARGMAX_BLOCKS + lane]; bi = cand_i[r * ARGMAX_BLOCKS + lane]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        *b.tok[r] = bi; toks_out[r] = bi;
        if (b.append[r]) *b.append[r] = bi;
    }
}
hipError_t launch_sample_batch(const float* logits, int V, const SampleBatch& b, int n, int64_t* toks_out_dev, void* scratch, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > MMD_ROUND_MAX_SAMPLERS) return hipErrorInvalidValue;
    float* cand_v = (float*)scratch; int* cand_i = (int*)((char*)scratch + (size_t)MMD_ROUND_MAX_SAMPLERS * ARGMAX_BLOCKS * sizeof(float));
    hipLaunchKernelGGL(argmax_penalty_multi_kernel, dim3(ARGMAX_BLOCKS, n), dim3(256), 0, st, logits, V, b, cand_v, cand_i);
    hipLaunchKernelGGL(argmax_final_multi_kernel, dim3(n), dim3(64), 0, st, cand_v, cand_i, b, toks_out_dev);
    return hipGetLastError();
}
size_t sample_batch_scratch_bytes() { return (size_t)MMD_ROUND_MAX_SAMPLERS * ARGMAX_BLOCKS * (sizeof(float) + sizeof(int)); }

// rows of a step that are the embedding of the token a sampler drew last (the feed rows of mmd_round_multi): out[row[r], :] = table[*tok[r], :]
template <typename T>
__global__ void embed_feed_kernel(const T* __restrict__ table, FeedBatch f, int H, long long vocab, T* __restrict__ out) {
    long long id = *f.tok[blockIdx.x];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const T* src = table + id * H;
    T* dst = out + (long long)f.row[blockIdx.x] * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) dst[c] = src[c];
}
hipError_t launch_embed_feed(int dtype, const void* table, const FeedBatch& f, int n, int H, int64_t vocab, void* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > MMD_ROUND_MAX_SAMPLERS) return hipErrorInvalidValue;
    if (dtype == MMD_F32) hipLaunchKernelGGL(embed_feed_kernel<float>, dim3(n), dim3(256), 0, st, (const float*)table, f, H, (long long)vocab, (float*)out);
    else hipLaunchKernelGGL(embed_feed_kernel<bf16_t>, dim3(n), dim3(256), 0, st, (const bf16_t*)table, f, H, (long long)vocab, (bf16_t*)out);
    return hipGetLastError();
}

// last node of the captured decode step: append the sampled token to the penalty list (unless EOS) and advance the position
__global__ void advance_state_kernel(StepState* st, const int64_t* __restrict__ tok, int64_t* __restrict__ prev, int prev_cap, long long eos, int use_penalty) {
    if (threadIdx.x != 0) return;
    const long long t = *tok;
    if (use_penalty && t != eos && st->n_prev < prev_cap) { prev[st->n_prev] = t; st->n_prev += 1; }
    st->n_ctx += 1;
}
hipError_t launch_advance_state(StepState* st_dev, const int64_t* tok_dev, int64_t* prev_dev, int prev_cap, int64_t eos, int use_penalty, hipStream_t st) {
    hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(64), 0, st, st_dev, tok_dev, prev_dev, prev_cap, (long long)eos, use_penalty);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// layout helpers used at weight-load time
// ---------------------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void convert_kernel(const S* __restrict__ s, D* __restrict__ d, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = from_f<D>(to_f<S>(s[i]));
}
hipError_t launch_convert(const void* src, int sdt, void* dst, int ddt, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
#define CV(S, D) hipLaunchKernelGGL((convert_kernel<S, D>), dim3(blocks), dim3(256), 0, st, (const S*)src, (D*)dst, (long long)n)
    if (sdt == MMD_F16) { if (ddt == MMD_F32) CV(f16_t, float); else if (ddt == MMD_F16) CV(f16_t, f16_t); else CV(f16_t, bf16_t); }
    else if (ddt == MMD_F16) { if (sdt == MMD_F32) CV(float, f16_t); else CV(bf16_t, f16_t); }
    else if (sdt == MMD_F32 && ddt == MMD_F32) CV(float, float);
    else if (sdt == MMD_F32) CV(float, bf16_t);
    else if (ddt == MMD_F32) CV(bf16_t, float);
    else CV(bf16_t, bf16_t);
#undef CV
    return hipGetLastError();
}

template <typename T>
__global__ void copy_rows_kernel(const T* __restrict__ s, long long lds_, T* __restrict__ d, long long ldd, int rows, int cols) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    d[(long long)r * ldd + c] = s[(long long)r * lds_ + c];
}
hipError_t launch_copy_rows(int dtype, const void* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols, hipStream_t st) {
    long long total = (long long)rows * cols;
    if (total <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)src, (long long)lds_, (float*)dst, (long long)ldd, rows, cols);
    else hipLaunchKernelGGL(copy_rows_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)src, (long long)lds_, (bf16_t*)dst, (long long)ldd, rows, cols);
    return hipGetLastError();
}

// out rows: [a 0..15, b 0..15, a 16..31, b 16..31, ...]  (gate/up interleave for the SwiGLU epilogue)
template <typename T>
__global__ void interleave16_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int rows, int cols) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2LL * rows * cols) return;
    long long orow = i / cols; int c = (int)(i % cols);
    long long blk = orow / 32; int within = (int)(orow % 32);
    long long srow = blk * 16 + (within & 15);
    const T* src = within < 16 ? a : b;
    out[i] = srow < rows ? src[srow * cols + c] : (T)0;
}
hipError_t launch_interleave16(int dtype, const void* a, const void* b, void* out, int rows, int cols, hipStream_t st) {
    long long total = 2LL * rows * cols;
    if (total <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(interleave16_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, rows, cols);
    else hipLaunchKernelGGL(interleave16_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, rows, cols);
    return hipGetLastError();
}

template <typename T>
__global__ void pad_cols_kernel(const T* __restrict__ s, int rows, int cols, T* __restrict__ d, int cols_pad) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cols_pad) return;
    int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
    d[i] = c < cols ? s[(long long)r * cols + c] : (T)0;
}
hipError_t launch_pad_cols(int dtype, const void* src, int rows, int cols, void* dst, int cols_pad, hipStream_t st) {
    long long total = (long long)rows * cols_pad;
    if (total <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(pad_cols_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)src, rows, cols, (float*)dst, cols_pad);
    else hipLaunchKernelGGL(pad_cols_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)src, rows, cols, (bf16_t*)dst, cols_pad);
    return hipGetLastError();
}

// W[out,in] += scale * B[out,r] . A[r,in] in fp32 (peft LoRA merge; models/modeling_live.py:123)
template <typename T>
__global__ void lora_merge_kernel(T* __restrict__ W, const float* __restrict__ A, const float* __restrict__ B, int out_f, int in_f, int r, float scale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)out_f * in_f) return;
    int o = (int)(i / in_f), c = (int)(i % in_f);
    float acc = 0.f;
    for (int k = 0; k < r; ++k) acc += B[(long long)o * r + k] * A[(long long)k * in_f + c];
    W[i] = from_f<T>(to_f<T>(W[i]) + scale * acc);
}
hipError_t launch_lora_merge(int dtype, void* W, const float* A, const float* B, int out_f, int in_f, int r, float scale, hipStream_t st) {
    long long total = (long long)out_f * in_f;
    if (dtype == MMD_F32) hipLaunchKernelGGL(lora_merge_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (float*)W, A, B, out_f, in_f, r, scale);
    else hipLaunchKernelGGL(lora_merge_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (bf16_t*)W, A, B, out_f, in_f, r, scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// image preprocess -- LLaVA SigLipImageProcessor.preprocess (test/inference.py:203): Pillow's 8-bit separable bicubic
// resampler (horizontal pass to a uint8 temp, vertical pass; fixed-point taps with PRECISION_BITS = 22, both passes
// round with 1 << 21 then clip8), then x * (1/255), (x - 0.5) / 0.5.  Bit-exact with Pillow; tap tables are built on
// the host (model.cpp) exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t clip8_fix(int v) { v >>= 22; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ void resize_h_kernel(const uint8_t* __restrict__ in, int R, int size, const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                int ksize, uint8_t* __restrict__ tmp, long long planes) {
    // in [planes][R][R] -> tmp [planes][R][size]
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * R * size) return;
    int xx = (int)(i % size); long long row = i / size;
    int xmin = bounds[2 * xx], xn = bounds[2 * xx + 1];
    const uint8_t* src = in + row * R + xmin;
    const int32_t* k = coef + (long long)xx * ksize;
    int acc = 1 << 21;
    for (int x = 0; x < xn; ++x) acc += (int)src[x] * k[x];
    tmp[i] = clip8_fix(acc);
}
template <typename T>
__global__ void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int R, int size, const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                     int ksize, T* __restrict__ out, long long planes, int identity) {
    // tmp [planes][R][size] -> out [planes][size][size], normalised
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * size * size) return;
    int xx = (int)(i % size); int yy = (int)((i / size) % size); long long pl = i / ((long long)size * size);
    uint8_t px;
    if (identity) px = tmp[(pl * R + yy) * size + xx];
    else {
        int ymin = bounds[2 * yy], yn = bounds[2 * yy + 1];
        const int32_t* k = coef + (long long)yy * ksize;
        int acc = 1 << 21;
        for (int y = 0; y < yn; ++y) acc += (int)tmp[(pl * R + ymin + y) * size + xx] * k[y];
        px = clip8_fix(acc);
    }
    float f = (float)px * (1.0f / 255.0f);          // HF rescale: image.astype(float32) * float32(1/255)
    f = (f - 0.5f) / 0.5f;
    out[i] = from_f<T>(f);
}
hipError_t launch_preprocess(int dtype, const uint8_t* frames, int T_, int R, int size, const int32_t* coef, const int32_t* bounds, int ksize,
                             uint8_t* tmp, void* out, hipStream_t st) {
    long long planes = (long long)T_ * 3;
    if (planes <= 0) return hipSuccess;
    int identity = (R == size);
    const uint8_t* vsrc = frames;
    if (!identity) {
        long long n1 = planes * R * size;
        hipLaunchKernelGGL(resize_h_kernel, dim3(cdiv(n1, 256)), dim3(256), 0, st, frames, R, size, coef, bounds, ksize, tmp, planes);
        vsrc = tmp;
    }
    long long n2 = planes * size * size;
    if (dtype == MMD_F32) hipLaunchKernelGGL(resize_v_norm_kernel<float>, dim3(cdiv(n2, 256)), dim3(256), 0, st, vsrc, R, size, coef, bounds, ksize, (float*)out, planes, identity);
    else hipLaunchKernelGGL(resize_v_norm_kernel<bf16_t>, dim3(cdiv(n2, 256)), dim3(256), 0, st, vsrc, R, size, coef, bounds, ksize, (bf16_t*)out, planes, identity);
    return hipGetLastError();
}


// the last preprocess pass fused into the patch-embed load.  Same arithmetic as resize_v_norm_kernel (vertical Pillow pass,
// 1/255, (x - .5)/.5, rounding to the model dtype), but the result is written straight into the im2col matrix of the patch-embed GEMM
// ([B*grid*grid][Kpad], k = (channel, ky, kx)) instead of a [B,3,size,size] pixel_values tensor that im2col would read back.
template <typename T>
__global__ void resize_v_norm_im2col_kernel(const uint8_t* __restrict__ tmp, int R, int size, const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                            int ksize, int identity, int patch, int grid, int Kpad, T* __restrict__ out) {
    const int row = blockIdx.x;                        // b*grid*grid + gy*grid + gx
    const int b = row / (grid * grid), rem = row % (grid * grid), gy = rem / grid, gx = rem % grid;
    const int K = 3 * patch * patch;
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
        T v = 0;
        if (k < K) {
            const int c = k / (patch * patch), r2 = k % (patch * patch), yy = gy * patch + r2 / patch, xx = gx * patch + r2 % patch;
            const long long pl = (long long)b * 3 + c;
            uint8_t px;
            if (identity) px = tmp[(pl * R + yy) * size + xx];
            else {
                const int ymin = bounds[2 * yy], yn = bounds[2 * yy + 1];
                const int32_t* kk = coef + (long long)yy * ksize;
                int acc = 1 << 21;
                for (int y = 0; y < yn; ++y) acc += (int)tmp[(pl * R + ymin + y) * size + xx] * kk[y];
                px = clip8_fix(acc);
            }
            float f = (float)px * (1.0f / 255.0f);
            f = (f - 0.5f) / 0.5f;
            if constexpr (std::is_same<T, f16_t>::value) f = bf2f(f2bf(f));          // the driver hands the tower bf16 pixel_values (test/inference.py:203); autocast then casts those to half
            v = from_f<T>(f);
        }
        out[(long long)row * Kpad + k] = v;
    }
}
hipError_t launch_preprocess_im2col(int dtype, const uint8_t* frames, int T_, int R, int size, const int32_t* coef, const int32_t* bounds, int ksize,
                                    uint8_t* tmp, int patch, int grid, int Kpad, void* out, hipStream_t st) {
    const long long planes = (long long)T_ * 3;
    if (planes <= 0) return hipSuccess;
    const int identity = (R == size);
    const uint8_t* vsrc = frames;
    if (!identity) {
        const long long n1 = planes * R * size;
        hipLaunchKernelGGL(resize_h_kernel, dim3(cdiv(n1, 256)), dim3(256), 0, st, frames, R, size, coef, bounds, ksize, tmp, planes);
        vsrc = tmp;
    }
    const int rows = T_ * grid * grid;
    if (dtype == MMD_F32) hipLaunchKernelGGL(resize_v_norm_im2col_kernel<float>, dim3(rows), dim3(256), 0, st, vsrc, R, size, coef, bounds, ksize, identity, patch, grid, Kpad, (float*)out);
    else if (dtype == MMD_F16) hipLaunchKernelGGL(resize_v_norm_im2col_kernel<f16_t>, dim3(rows), dim3(256), 0, st, vsrc, R, size, coef, bounds, ksize, identity, patch, grid, Kpad, (f16_t*)out);
    else hipLaunchKernelGGL(resize_v_norm_im2col_kernel<bf16_t>, dim3(rows), dim3(256), 0, st, vsrc, R, size, coef, bounds, ksize, identity, patch, grid, Kpad, (bf16_t*)out);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// letterbox -- the resize + pad + channel flip of the reference's load_video (test/datasets.py:52-71,
// demo/liveinfer.py:32-51): cv2.resize(frame, (new_w, new_h)) [INTER_LINEAR, 8-bit], cv2.copyMakeBorder(constant),
// cv2.cvtColor(BGR2RGB), HWC -> CHW.  One thread per output pixel; the resize is OpenCV's published 8-bit fixed-point
// bilinear (imgproc/resize.cpp: 11-bit tap weights as shorts, horizontal pass kept at 2^11 scale in int, vertical pass
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2), and its exact-2x shortcut (INTER_LINEAR -> 2x2 INTER_AREA,
// (a+b+c+d+2)>>2).  Tap tables (xofs/alpha, yofs/beta) are built on the host with OpenCV's float arithmetic (model.hip).
// src [T][H][W][3] uint8, dst [T][3][R][R] uint8.
// ---------------------------------------------------------------------------------------------------------------
__global__ void letterbox_kernel(const uint8_t* __restrict__ src, int H, int W, int new_w, int new_h, int R, int top, int left,
                                 const int32_t* __restrict__ xtab, const int32_t* __restrict__ ytab, int area2x, int flip,
                                 uint32_t pad, uint8_t* __restrict__ dst, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % R), y = (int)((i / R) % R);
    const long long t = i / ((long long)R * R);
    const int rx = x - left, ry = y - top;
    uint8_t o[3];
    if (rx < 0 || rx >= new_w || ry < 0 || ry >= new_h) {
        o[0] = pad & 255; o[1] = (pad >> 8) & 255; o[2] = (pad >> 16) & 255;
    } else {
        const uint8_t* f = src + t * (long long)H * W * 3;
        if (area2x) {
            const uint8_t* r0 = f + ((long long)(2 * ry) * W + 2 * rx) * 3;
            const uint8_t* r1 = r0 + (long long)W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (uint8_t)(((int)r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2);
        } else {
            const int sx0 = xtab[4 * rx], sx1 = xtab[4 * rx + 1], a0 = xtab[4 * rx + 2], a1 = xtab[4 * rx + 3];
            const int sy0 = ytab[4 * ry], sy1 = ytab[4 * ry + 1], b0 = ytab[4 * ry + 2], b1 = ytab[4 * ry + 3];
            const uint8_t* r0 = f + (long long)sy0 * W * 3;
            const uint8_t* r1 = f + (long long)sy1 * W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = (int)r0[sx0 * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;
                const int h1 = (int)r1[sx0 * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
                const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    uint8_t* d = dst + t * 3LL * R * R + (long long)y * R + x;
    const long long plane = (long long)R * R;
    d[0] = o[flip ? 2 : 0]; d[plane] = o[1]; d[2 * plane] = o[flip ? 0 : 2];
}
hipError_t launch_letterbox(const uint8_t* src, int T_, int H, int W, int new_w, int new_h, int R, int top, int left, const int32_t* xtab,
                            const int32_t* ytab, int area2x, int flip, uint32_t pad, uint8_t* dst, hipStream_t st) {
    long long total = (long long)T_ * R * R;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(letterbox_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, src, H, W, new_w, new_h, R, top, left, xtab, ytab, area2x, flip, pad, dst, total);
    return hipGetLastError();
}

// ---- secondary encoders (models/vision_live.py: HF SigLIP-L/16-384 and CLIP-L/14-336 towers) -----------------------------------------
// torchvision normalize(frames * rescale, mean, std) (models/vision_live.py:13,36): uint8 or float [B,3,R,R] -> ctx dtype
template <typename S, typename T>
__global__ void normalize_frames_kernel(const S* __restrict__ in, long long per_ch, long long total, float rescale, float m0, float m1, float m2, float s0, float s1, float s2,
                                        T* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)((i / per_ch) % 3);
    const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2), sd = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
    out[i] = from_f<T>(((float)in[i] * rescale - mean) / sd);
}
hipError_t launch_normalize_frames(int dtype, const void* in, int in_kind, int B, int R, float rescale, const float* mean, const float* sd, void* out, hipStream_t st) {
    const long long per_ch = (long long)R * R, total = (long long)B * 3 * per_ch;
    if (total <= 0) return hipSuccess;
    dim3 grid(cdiv(total, 256)), block(256);
#define NF(S, T) hipLaunchKernelGGL((normalize_frames_kernel<S, T>), grid, block, 0, st, (const S*)in, per_ch, total, rescale, mean[0], mean[1], mean[2], sd[0], sd[1], sd[2], (T*)out)
    if (dtype == MMD_F32) { if (in_kind == 0) NF(uint8_t, float); else NF(float, float); }
    else { if (in_kind == 0) NF(uint8_t, bf16_t); else NF(float, bf16_t); }
#undef NF
    return hipGetLastError();
}
// CLIPVisionEmbeddings.forward (clip/modeling_clip.py [3P]): cat(class_embedding, patch_embeds) + position_embedding; patch [B*T, C] -> out [B, 1+T, C]
template <typename T>
__global__ void assemble_cls_kernel(const T* __restrict__ patch, const T* __restrict__ cls, const T* __restrict__ pos, int Tk, int C, long long total, T* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long long row = i / C;
    const int t = (int)(row % (Tk + 1));
    const long long b = row / (Tk + 1);
    const float v = t == 0 ? to_f<T>(cls[c]) : to_f<T>(patch[(b * Tk + (t - 1)) * C + c]);
    out[i] = from_f<T>(v + to_f<T>(pos[(long long)t * C + c]));
}
hipError_t launch_assemble_cls(int dtype, const void* patch, const void* cls, const void* pos, int B, int Tk, int C, void* out, hipStream_t st) {
    const long long total = (long long)B * (Tk + 1) * C;
    if (total <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(assemble_cls_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)patch, (const float*)cls, (const float*)pos, Tk, C, total, (float*)out);
    else hipLaunchKernelGGL(assemble_cls_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos, Tk, C, total, (bf16_t*)out);
    return hipGetLastError();
}
// quick_gelu (CLIP): x * sigmoid(1.702 x), in place on the (storage-rounded) fc1 output
template <typename T>
__global__ void quick_gelu_kernel(T* __restrict__ x, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = to_f<T>(x[i]);
    x[i] = from_f<T>(v * (1.0f / (1.0f + __expf(-1.702f * v))));
}
hipError_t launch_quick_gelu(int dtype, void* x, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (dtype == MMD_F32) hipLaunchKernelGGL(quick_gelu_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, (float*)x, (long long)n);
    else hipLaunchKernelGGL(quick_gelu_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, (bf16_t*)x, (long long)n);
    return hipGetLastError();
}
