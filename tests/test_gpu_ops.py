"""Parity of the HIP operators (called through the C ABI) against the oracle / plain fp32 torch math.
Tolerances: fp32 kernels 1e-4-class (accumulation order only); bf16 kernels are compared with the same math on
bf16-rounded inputs and may differ by bf16 output rounding (2^-8 relative) plus accumulation order."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
from oracle import duet_oracle as O
from conftest import load_npz

DT = [torch.float32, torch.bfloat16]


@pytest.fixture(scope='module', params=DT, ids=['f32', 'bf16'])
def ops(request):
    from rawops import RawOps
    return RawOps(request.param)


def rt(x, dtype):          # round-trip through the storage type
    return x.to(dtype).float()


def assert_close(got, ref, dtype, scale=1.0, what=''):
    got = got.float().cpu(); ref = ref.float().cpu()
    tol = (2e-5 if dtype == torch.float32 else 1.2e-2) * scale
    err = (got - ref).abs().max().item()
    den = ref.abs().max().item() + 1e-6
    assert err <= tol * max(1.0, den), f'{what}: max abs err {err:.3e} (ref max {den:.3e}, tol {tol:.1e})'


GEMM_SHAPES = [(49, 64, 64), (7, 130, 72), (200, 96, 588), (64, 256, 512), (300, 384, 256), (1, 512, 64), (129, 160, 4352), (392, 512, 1152), (729, 256, 640), (1000, 1152, 1152), (520, 96, 128), (200, 64, 64), (130, 128, 192), (257, 160, 320), (300, 96, 448)]


@pytest.fixture(scope='module')
def ops_bf16():
    from rawops import RawOps
    return RawOps(torch.bfloat16)


GENERIC_VARIANTS = [1, 2, 3]                                              # both dtypes, any shape
TILE_VARIANTS = [4, 6, 7, 32, 33, 36, 37]     # bf16 MFMA tile kernels with shape conditions (16 + ring flags: 16 / 17 = the shipped 8- / 4-wave instantiations, + 4 = split K)


def _applies(variant, M, N, K):
    if variant == 4:            # big-tile kernel
        return M > 64 and N % 64 == 0 and K % 64 == 0
    return M > 64 and N % 32 == 0 and K % 64 == 0      # 256x256 ring family


def _gemm_bias(ops, M, N, K, variant):
    g = torch.Generator().manual_seed(M * 7 + N)
    X = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
    Y = ops.gemm(X, W, b, variant=variant)
    ref = F.linear(rt(X, ops.dtype), rt(W, ops.dtype), rt(b, ops.dtype))
    assert_close(Y, ref, ops.dtype, what=f'gemm {M}x{N}x{K} v{variant}')


@pytest.mark.parametrize('M,N,K', GEMM_SHAPES)
@pytest.mark.parametrize('variant', GENERIC_VARIANTS)
def test_gemm_bias(ops, M, N, K, variant):
    _gemm_bias(ops, M, N, K, variant)


@pytest.mark.parametrize('M,N,K,variant', [(m, n, k, v) for v in TILE_VARIANTS for (m, n, k) in GEMM_SHAPES if _applies(v, m, n, k)])
def test_gemm_bias_tile_kernels(ops_bf16, M, N, K, variant):
    _gemm_bias(ops_bf16, M, N, K, variant)


@pytest.mark.parametrize('epi', ['gelu_tanh', 'gelu_erf', 'resid', 'swiglu'])
@pytest.mark.parametrize('variant', GENERIC_VARIANTS)
def test_gemm_epilogues(ops, epi, variant):
    _gemm_epilogues(ops, epi, variant)


@pytest.mark.parametrize('epi', ['gelu_tanh', 'gelu_erf', 'resid', 'swiglu'])
@pytest.mark.parametrize('variant', TILE_VARIANTS)
def test_gemm_epilogues_tile_kernels(ops_bf16, epi, variant):
    _gemm_epilogues(ops_bf16, epi, variant)


def _gemm_epilogues(ops, epi, variant):
    g = torch.Generator().manual_seed(11)
    M, N, K = (70, 192, 136) if variant in GENERIC_VARIANTS else (300, 320, 192)
    assert variant in GENERIC_VARIANTS or _applies(variant, M, N, K)
    X = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / math.sqrt(K); b = 0.1 * torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    Xr, Wr, br, Rr = (rt(t, ops.dtype) for t in (X, W, b, R))
    if epi == 'swiglu':
        gate, up = Wr[:N // 2], Wr[N // 2:]
        # interleave rows in blocks of 16: [g0..15, u0..15, g16..31, ...] (the layout mmd_finalize_weights builds)
        Wi = torch.stack([gate.view(-1, 16, K), up.view(-1, 16, K)], 1).reshape(N, K)
        Y = ops.gemm(X, Wi, None, epi=epi, variant=variant)
        ref = F.silu(Xr @ gate.T) * (Xr @ up.T)
    elif epi == 'resid':
        Y = ops.gemm(X, W, b, R=R, epi=epi, variant=variant)
        ref = Rr + F.linear(Xr, Wr, br)
    else:
        Y = ops.gemm(X, W, b, epi=epi, variant=variant)
        lin = F.linear(Xr, Wr, br)
        ref = O.gelu_tanh(lin) if epi == 'gelu_tanh' else O.gelu_erf(lin)
    assert_close(Y, ref, ops.dtype, scale=2.0, what=f'{epi} v{variant}')


def test_gemm_fp32_output_is_rounded_like_reference(ops):
    """lm_head: the reference computes in the model dtype and then `.float()` (video_head_live_llava_qwen.py:155)."""
    g = torch.Generator().manual_seed(5)
    X = torch.randn(3, 64, generator=g); W = torch.randn(100, 64, generator=g) / 8
    Y = ops.gemm(X, W, out_f32=True)
    assert Y.dtype == torch.float32
    assert torch.equal(Y.cpu(), Y.cpu().to(ops.dtype).float())
    assert_close(Y, rt(X, ops.dtype) @ rt(W, ops.dtype).T, ops.dtype)


def test_norms(ops):
    g = torch.Generator().manual_seed(3)
    for M, H in ((5, 64), (49, 3584), (33, 1152), (2, 72)):
        x = torch.randn(M, H, generator=g) * 3; w = 1 + 0.1 * torch.randn(H, generator=g); b = 0.1 * torch.randn(H, generator=g)
        xr, wr, br = rt(x, ops.dtype).to(ops.dtype), rt(w, ops.dtype).to(ops.dtype), rt(b, ops.dtype).to(ops.dtype)
        assert_close(ops.rmsnorm(x, w, 1e-6), O.rms_norm(xr, wr, 1e-6), ops.dtype, what='rmsnorm')
        assert_close(ops.layernorm(x, w, b, 1e-6), O.layer_norm(xr, wr, br, 1e-6), ops.dtype, what='layernorm')


@pytest.mark.parametrize('M,H,period', [(5, 1152, 0), (729 * 3, 1152, 729), (37, 64, 0), (130, 2048, 13), (1, 72, 0)])
def test_resid32_layernorm_is_the_autocast_residual_step(M, H, period):
    """resid32_layernorm_kernel (the fp16 tower's fp32 residual stream, models/modeling_live.py:28 under autocast): h32 += float(y16) (or = position row + y16 for the
    embeddings), LayerNorm of the fp32 row -> fp16, and the bf16 image of the row for the tower's result -- against the same steps in torch: the fp32 stream and the bf16
    image BIT-exact (one fp32 add per element, one rounding), the LayerNorm output to fp16 rounding of the fp32 result."""
    import ctypes as C
    from rawops import RawOps, _ptr
    from mmduet_amd._lib import lib, check
    ops = RawOps(torch.bfloat16)
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(M + H)
    y = (torch.randn(M, H, generator=g, device=dev) * 2).to(torch.float16)
    h = torch.randn(M, H, generator=g, device=dev) * 5
    w = (1 + 0.1 * torch.randn(H, generator=g, device=dev)).to(torch.float16); b = (0.1 * torch.randn(H, generator=g, device=dev)).to(torch.float16)
    pos = (torch.randn(max(period, 1), H, generator=g, device=dev) * 0.5).to(torch.float16) if period else None
    ops.m._bind_stream()
    # (1) add + LayerNorm, output in place of y
    h1, y1 = h.clone(), y.clone()
    check(lib().mmd_op_resid32_layernorm(ops.ctx, _ptr(y1), _ptr(h1), _ptr(pos), period, _ptr(w), _ptr(b), _ptr(y1), None, M, H, 1e-6), ops.ctx)
    torch.cuda.synchronize()
    ref_h = (pos.float()[torch.arange(M, device=dev) % period] if period else h) + y.float()
    assert torch.equal(h1, ref_h)
    ref_ln = torch.nn.functional.layer_norm(ref_h.double(), (H,), w.double(), b.double(), 1e-6)
    err = (y1.double() - ref_ln).abs().max().item()
    assert err <= 2.0 ** -10 * max(1.0, ref_ln.abs().max().item()), (M, H, err)
    # (2) add only, bf16 image out, the fp32 stream left alone
    h2 = h.clone(); out = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
    check(lib().mmd_op_resid32_layernorm(ops.ctx, _ptr(y), _ptr(h2), _ptr(pos), period, None, None, None, _ptr(out), M, H, 1e-6), ops.ctx)
    torch.cuda.synchronize()
    assert torch.equal(out, ref_h.to(torch.bfloat16)) and torch.equal(h2, h)
    # operand checks
    assert lib().mmd_op_resid32_layernorm(ops.ctx, _ptr(y), _ptr(h2), None, 0, _ptr(w), None, None, None, M, H, 1e-6) != 0


@pytest.mark.parametrize('pos0', [0, 1000, 30000])
def test_rope_and_kv_append(ops, pos0):
    g = torch.Generator().manual_seed(pos0 + 1)
    S, nh, nkv, d, theta, cap = 9, 4, 2, 32, 1e6, 30080
    qkv = torch.randn(S, (nh + 2 * nkv) * d, generator=g)
    Kc = torch.zeros(nkv, cap, d, device=ops.dev, dtype=ops.dtype); Vc = torch.zeros_like(Kc)
    q = ops.rope_append(qkv, nh, nkv, d, theta, pos0, Kc, Vc)
    x = rt(qkv, ops.dtype).to(ops.dtype)
    cfg = O.OracleConfig(hidden_size=nh * d, num_attention_heads=nh, num_key_value_heads=nkv, rope_theta=theta)
    cos, sin = O.rope_tables(cfg, torch.arange(pos0, pos0 + S), ops.dtype)
    qr = O.apply_rope(x[:, :nh * d].view(S, nh, d).transpose(0, 1), cos, sin).transpose(0, 1).reshape(S, nh * d)
    kr = O.apply_rope(x[:, nh * d:(nh + nkv) * d].view(S, nkv, d).transpose(0, 1), cos, sin)
    vr = x[:, (nh + nkv) * d:].view(S, nkv, d).transpose(0, 1)
    # the raw entry point builds theta^(-2i/d) in double; torch's fp32 pow differs by <= 1 ulp, which the angle
    # pos * inv_freq amplifies by pos (the model path takes torch's table via mmd_set_rope_inv_freq and is exact)
    sc = 1.0 + pos0 * 2e-3
    assert_close(q, qr, ops.dtype, scale=sc if ops.dtype == torch.float32 else 1.0, what='rope q')
    assert_close(Kc[:, pos0:pos0 + S], kr, ops.dtype, scale=sc if ops.dtype == torch.float32 else 1.0, what='rope k')
    assert torch.equal(Vc[:, pos0:pos0 + S].cpu(), vr)
    assert Kc[:, :pos0].abs().sum() == 0 and Kc[:, pos0 + S:].abs().sum() == 0


def ref_attention(q, K, V, nh, nkv, d, n_ctx, causal, dtype):
    """Oracle attention math (oracle.duet_oracle.llm_layer) on [S, nh*d] / [nkv, n, d]."""
    S = q.shape[0]; n_tot = n_ctx + S
    qh = q.view(S, nh, d).transpose(0, 1).float()
    rep = nh // nkv
    kk = K[:, :n_tot].float()[:, None].expand(nkv, rep, n_tot, d).reshape(nh, n_tot, d)
    vv = V[:, :n_tot].float()[:, None].expand(nkv, rep, n_tot, d).reshape(nh, n_tot, d)
    s = qh @ kk.transpose(1, 2) * d ** -0.5
    if causal:
        mask = torch.arange(n_tot)[None, :] > (torch.arange(S)[:, None] + n_ctx)
        s = s.masked_fill(mask[None], float('-inf'))
    a = torch.softmax(s, -1)
    return (a @ vv).transpose(0, 1).reshape(S, nh * d)


ATTN_CASES = [  # S, nh, nkv, d, n_ctx, causal
    (1, 4, 2, 16, 0, True), (7, 4, 2, 16, 5, True), (49, 4, 1, 32, 300, True), (130, 4, 2, 16, 41, True),
    (49, 28, 4, 128, 0, True), (49, 28, 4, 128, 3000, True), (1, 28, 4, 128, 2500, True), (3, 28, 4, 128, 70, True), (98, 28, 4, 128, 777, True),
    (49, 28, 4, 128, 15000, True), (20, 8, 8, 128, 100, False), (729, 16, 16, 72, 0, False), (300, 4, 4, 72, 0, False), (130, 2, 1, 24, 0, False), (200, 8, 8, 72, 0, False),
    (16, 2, 2, 24, 0, False), (33, 4, 4, 64, 100, False), (577, 16, 16, 64, 0, False), (257, 12, 12, 64, 0, False), (196, 16, 16, 72, 533, False), (64, 3, 3, 64, 0, False),
]


@pytest.mark.parametrize('S,nh,nkv,d,n_ctx,causal', ATTN_CASES)
def test_attention_generic_kernel(ops, S, nh, nkv, d, n_ctx, causal):
    _attention(ops, S, nh, nkv, d, n_ctx, causal, 1)


# the MFMA attention kernels are bf16 only; variant 3 (the GQA flash kernel) is specialised for head_dim 128
@pytest.mark.parametrize('S,nh,nkv,d,n_ctx,causal,variant', [c + (v,) for v in (2, 3, 4) for c in ATTN_CASES if v != 3 or c[3] == 128])
def test_attention_mfma_kernels(ops_bf16, S, nh, nkv, d, n_ctx, causal, variant):
    _attention(ops_bf16, S, nh, nkv, d, n_ctx, causal, variant)


# decode geometry (<= 16 rows per kv head): the loader / compute ring of attn_gqa128_kernel<1, 4> -- 1..5+ tiles per split, ring refills, partial last tiles, empty context
@pytest.mark.parametrize('S,n_ctx', [(s, n) for s in (1, 2) for n in (0, 1, 62, 63, 64, 127, 191, 255, 256, 1000, 4100, 16383, 16400, 33000, 70001)])
def test_decode_attention_ring_contexts(ops_bf16, S, n_ctx):
    _attention(ops_bf16, S, 28, 4, 128, n_ctx, True, 3)


def _attention(ops, S, nh, nkv, d, n_ctx, causal, variant):
    g = torch.Generator().manual_seed(S * 13 + d)
    cap = (n_ctx + S + 37 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g); K = torch.randn(nkv, cap, d, generator=g); V = torch.randn(nkv, cap, d, generator=g)
    K[:, n_ctx + S:] = 1e4; V[:, n_ctx + S:] = 1e4          # poison beyond the valid range: must never be read into the result
    Kc, Vc = K.to(ops.dev, ops.dtype), V.to(ops.dev, ops.dtype)
    o = ops.attention(q, Kc, Vc, nh, nkv, d, n_ctx, causal, variant)
    ref = ref_attention(rt(q, ops.dtype), rt(K, ops.dtype), rt(V, ops.dtype), nh, nkv, d, n_ctx, causal, ops.dtype)
    assert torch.isfinite(o.float()).all()
    assert_close(o, ref, ops.dtype, scale=1.5, what=f'attention v{variant}')


def test_vit_ring_attention_equals_register_staged_kernel_at_op_level():
    """attn_d72_ring_kernel against attn_rowmajor_kernel<3, 5> (MMDUET_VIT_ATTN_RING=0) through the raw attention op, bf16: the tower's shapes (SigLIP 729 x 16 x 72, its
    196-row last layer over 729 keys) and ragged ends (1 .. 64 keys in the last tile, fewer rows than a block, many heads).  Same products; since round 6 the ring kernel
    adds the dims 64..71 part of a score with one fp32 v_add behind the 64-dim MFMA chain (D72_TAIL_SEPARATE: no MFMA reads a different-depth MFMA's result) where the
    register-staged kernel accumulates it inside the chain: the same fp32 terms in another order -> outputs equal to one bf16 ulp on a few elements, and the ring kernel
    itself bit-reproducible.  (head_dim 64 -- the secondary towers -- stays on attn_rowmajor_kernel<2, 4>: it already runs four blocks per CU, the ring form
    measured 1755-1782 against 1786-1787 frames/s on the native-336 line.)"""
    import subprocess, sys, os, json
    from conftest import ROOT
    code = r'''
import os, sys, json, hashlib, torch
sys.path.insert(0, os.environ["MMD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MMD_ROOT"], "tests"))
from rawops import RawOps
ops = RawOps(torch.bfloat16)
res = {}
for S, nh, d, n_ctx in ((729, 16, 72, 0), (196, 16, 72, 533), (577, 24, 72, 0), (257, 12, 72, 0), (65, 2, 72, 0), (64, 2, 72, 1), (130, 3, 72, 63), (70, 2, 72, 58)):
    g = torch.Generator().manual_seed(S * 7 + d)
    cap = (n_ctx + S + 37 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g); K = torch.randn(nh, cap, d, generator=g); V = torch.randn(nh, cap, d, generator=g)
    o = ops.attention(q, K.to(ops.dev, ops.dtype), V.to(ops.dev, ops.dtype), nh, nh, d, n_ctx, False, 4)
    assert torch.isfinite(o.float()).all()
    res[f"{S}_{nh}_{d}_{n_ctx}"] = o.float().cpu().flatten().tolist()
print("RES " + json.dumps(res))
'''
    def run(**kw):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MMD_ROOT=ROOT, **kw), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('RES ')][0][4:])
    ring, staged = run(), run(MMDUET_VIT_ATTN_RING='0')
    assert len(ring) == 8
    for k in ring:
        a, b = torch.tensor(ring[k]), torch.tensor(staged[k])
        assert (a - b).abs().max().item() <= 2 ** -7 * max(1.0, b.abs().max().item()), (k, (a - b).abs().max().item())          # one bf16 ulp at the output's largest magnitude
        assert (a - b).abs().mean().item() <= 2 ** -11 * max(1.0, b.abs().max().item()), (k, (a - b).abs().mean().item())          # ... on few elements
    again = run()
    assert all(again[k] == ring[k] for k in ring)


def test_pooling_modes(ops):
    g = torch.Generator().manual_seed(2)
    for grid, stride in ((27, 4), (4, 2), (5, 2)):
        x = torch.randn(2, grid * grid, 40, generator=g)
        for mode, name in ((0, 'bilinear'), (1, 'average'), (2, 'max')):
            cfg = O.tiny_config(video_pooling_stride=stride, mm_spatial_pool_mode=name)
            ref = O.post_projector_pooling(cfg, rt(x, ops.dtype).to(ops.dtype))
            assert_close(ops.pool(x, grid, mode, stride), ref, ops.dtype, what=f'pool {name} {grid}/{stride}')


def test_adaptive_avg_pool_secondary_path(ops):
    """adaptive_avg_pool2d of the token grid to frame_token_pooled (models/vision_live.py:17-24)."""
    g = torch.Generator().manual_seed(4)
    for grid, out in ((24, 7), (27, 7), (5, 3), (4, 4)):
        x = torch.randn(2, grid * grid, 24, generator=g)
        ref = O.adaptive_avg_pool_tokens(rt(x, ops.dtype).to(ops.dtype), (out, out))
        assert_close(ops.pool(x, grid, 3, out), ref, ops.dtype, what=f'adaptive pool {grid}->{out}')


def test_preprocess_bit_exact_with_pillow(ops):
    """mmd_preprocess_frames against image_processor outputs recorded from the reference stack (PIL bicubic)."""
    z = load_npz('preprocess.npz')
    from helpers import hip_model
    m, _, _ = hip_model('A', ops.dtype)           # tower resolution 56
    for tag in ('same', 'up', 'down'):
        pv = m.get_vision_tower().image_processor.preprocess(torch.from_numpy(z[f'{tag}_frames']))['pixel_values']
        ref = torch.from_numpy(z[f'{tag}_pixel_values'])
        if ops.dtype == torch.float32:
            assert torch.equal(pv.cpu(), ref), (pv.cpu() - ref).abs().max()
        else:
            assert torch.equal(pv.cpu(), ref.to(torch.bfloat16))


# ---- gemm_stream_kernel: the weight-streaming GEMM of the 32 < M <= 256 regime (round 4) --------------------------------------------------------------
@pytest.mark.parametrize('M', [33, 49, 53, 64, 75, 98, 128, 147, 196, 245, 256])
@pytest.mark.parametrize('N,K', [(4608, 3584), (3584, 3584), (3584, 18944), (160, 256), (1152, 384)])
def test_stream_gemm_slabs_equal_fp32_math(M, N, K):
    """Slab mode (the fused LLM schedule's qkv / o_proj / down_proj at per-frame and short-chunk sizes): the fp32 partial slabs of gemm_stream_kernel must SUM to
    X . W^T in fp32 arithmetic -- tolerance 2e-5 x |ref|max x sqrt(K / 3584) (accumulation order only: the products are exact, nothing is rounded to bf16 here).
    Covers every instantiation (M <= 64 / 128 / 256), K splits with a shorter last block, n-tile tails (N = 160: 10 n-tiles over 4-wave groups) and rows beyond M."""
    import math
    from rawops import RawOps
    ops = RawOps(torch.bfloat16)
    g = torch.Generator(device=ops.dev).manual_seed(M * 7 + N + K)
    X = (torch.randn(M, K, generator=g, device=ops.dev) * 0.7).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=ops.dev) / math.sqrt(K)).to(torch.bfloat16)
    Y, n = ops.gemm_slabs(X, W, variant=8)
    import ctypes as C
    from mmduet_amd._lib import lib
    plan = (C.c_int * 4)(); lib().mmd_op_gemm_last_plan(ops.ctx, plan)
    assert plan[0] == 8, list(plan)                         # GEMM_K_STREAM really ran
    ref = X.double() @ W.double().T
    err = (Y.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert n >= 1 and err <= 2e-5 * max(1.0, math.sqrt(K / 3584)), (M, N, K, n, err)


@pytest.mark.parametrize('M', [33, 49, 64, 98, 128, 147, 196, 256])
def test_stream_gemm_swiglu_matches_fp32_and_is_deterministic(M):
    """gate_up at the true width through gemm_stream_kernel's SwiGLU epilogue (no K split): bf16 bound against fp32 math, identical bits on repetition."""
    import math
    import torch.nn.functional as F
    from rawops import RawOps
    ops = RawOps(torch.bfloat16)
    N, K = 37888, 3584
    g = torch.Generator(device=ops.dev).manual_seed(M)
    X = (torch.randn(M, K, generator=g, device=ops.dev) * 0.7).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=ops.dev) / math.sqrt(K)).to(torch.bfloat16)
    gate, up = W[:N // 2], W[N // 2:]
    Wi = torch.stack([gate.view(-1, 16, K), up.view(-1, 16, K)], 1).reshape(N, K).contiguous()
    Y = ops.gemm(X, Wi, None, epi='swiglu', variant=8)
    Xf = X.float()
    ref = F.silu((Xf @ gate.float().T).to(torch.bfloat16).float()).to(torch.bfloat16).float() * (Xf @ up.float().T).to(torch.bfloat16).float()
    err = (Y.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err <= 2.4e-2, err
    assert torch.equal(Y, ops.gemm(X, Wi, None, epi='swiglu', variant=8))


@pytest.mark.parametrize('M', [49, 75, 130, 256])
@pytest.mark.parametrize('N,K,epi', [(4608, 3584, 'none'), (3584, 18944, 'resid'), (1152, 1152, 'gelu_tanh')])
def test_stream_gemm_with_the_epilogue_in_place(M, N, K, epi):
    """The unfused form (several streams per forward, MMDUET_NO_FUSE): gemm_stream_kernel's slabs + the serial reduce with bias / residual / activation -- the bf16 bound
    against fp32 math, and the dispatcher really took the streaming kernel."""
    import math, ctypes as C
    import torch.nn.functional as F
    from oracle import duet_oracle as O
    from rawops import RawOps
    from mmduet_amd._lib import lib
    ops = RawOps(torch.bfloat16)
    g = torch.Generator(device=ops.dev).manual_seed(M + N)
    X = (torch.randn(M, K, generator=g, device=ops.dev) * 0.7).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=ops.dev) / math.sqrt(K)).to(torch.bfloat16)
    b = (0.1 * torch.randn(N, generator=g, device=ops.dev)).to(torch.bfloat16)
    R = torch.randn(M, N, generator=g, device=ops.dev).to(torch.bfloat16) if epi == 'resid' else None
    Y = ops.gemm(X, W, b, R=R, epi=epi, variant=0)
    plan = (C.c_int * 4)(); lib().mmd_op_gemm_last_plan(ops.ctx, plan)
    assert plan[0] == 8, list(plan)
    lin = F.linear(X.float(), W.float(), b.float())
    ref = lin.to(torch.bfloat16).float() + R.float() if epi == 'resid' else (O.gelu_tanh(lin.to(torch.bfloat16).float()) if epi == 'gelu_tanh' else lin)
    err = (Y.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err <= 2.4e-2, err
