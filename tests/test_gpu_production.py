"""Parity in the regimes the benchmark actually runs (VERDICT r01 "next round" item 1).

Every case goes through the C ABI and is compared with plain fp32 torch math / the oracle (oracle/duet_oracle.py is device-agnostic
torch code; here it runs on the GPU through torch's own fp32 kernels -- an independent implementation -- on the SAME bf16-rounded
weights and inputs).  Covered: the persistent 256-row ring GEMM with > 256 tiles (35-frame tower batch, M = 25 515), the 1274-row LLM
chunk incl. the automatic split-K over grid.z, chunk attention at S = 1274 over 0 / 15 k / 30 k keys (cost-model split-KV), the 35-frame
tower batch end to end, a 26-frame chunk against 26 per-frame steps, fp32 mode at the true widths, and one FULL-DEPTH (26 + 28 layers)
stream prefix with measured head-logit deltas.  Measured deltas are written to gpurun_out/parity_r02.json (copied to profiles/).

Tolerances (stated per test): bf16 kernels vs fp32 math on the same bf16 inputs -- output rounding (2^-9 relative) plus accumulation
order; fp32 mode -- 1e-3 on head logits (the north-star figure)."""
import json, math, os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
from oracle import duet_oracle as O
from conftest import ROOT

RESULTS = {}


def _record(key, **vals):
    RESULTS[key] = {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in vals.items()}
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'parity_r04.json')
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[key] = RESULTS[key]
        json.dump(cur, open(path, 'w'), indent=1, sort_keys=True)
    except Exception:
        pass


@pytest.fixture(scope='module')
def ops():
    from rawops import RawOps
    return RawOps(torch.bfloat16)


def _plan(ops):
    import ctypes as C
    from mmduet_amd._lib import lib
    p = (C.c_int * 4)()
    lib().mmd_op_gemm_last_plan(ops.ctx, p)
    return dict(kernel=p[0], tiles=p[1], splits=p[2], blocks=p[3])


RING = (6, 7)       # GEMM_K_RING256 / the 4-wave 256x128 ring


def _rel_err(got, ref):
    return (got.float() - ref.float()).abs().max().item() / max(1.0, ref.float().abs().max().item())


# ---- (a) production GEMM shapes through the dispatcher ---------------------------------------------------------------------------
PROD_GEMMS = [  # name, M, N, K, epilogue, bias, expects
    ('vit_qkv', 25515, 3456, 1152, 'none', True, dict(min_tiles=257)),
    ('vit_fc1', 25515, 4352, 1152, 'gelu_tanh', True, dict(min_tiles=257)),
    ('vit_fc2', 25515, 1152, 4352, 'resid', True, dict(min_tiles=257)),
    ('vit_o', 25515, 1152, 1152, 'resid', True, dict(min_tiles=257)),
    ('proj0', 25515, 3584, 1152, 'gelu_erf', True, dict(min_tiles=257)),
    ('llm_gate_up', 1274, 37888, 3584, 'swiglu', False, dict(min_tiles=257)),
    ('llm_down', 1274, 3584, 18944, 'resid', False, dict(min_splits=2)),
    ('llm_qkv', 1274, 4608, 3584, 'none', True, dict()),
    ('llm_o', 1274, 3584, 3584, 'resid', False, dict(tiles=448)),                       # 160-row tiles: 8 x 56 blocks, two per CU in one round (128-row tiles: 560)
    ('llm_o_13_frames', 637, 3584, 3584, 'resid', False, dict(tiles=224)),
    ('llm_o_tail', 1303, 3584, 3584, 'resid', True, dict(tiles=504)),                    # M % 160 != 0: masked rows of the last tile
    ('llm_gate_up_tail', 1303, 37888, 3584, 'swiglu', False, dict(min_tiles=257)),      # chunk + text prefix: M % 256 != 0 and M % 16 != 0
]


@pytest.mark.parametrize('name,M,N,K,epi,has_bias,expect', PROD_GEMMS, ids=[p[0] for p in PROD_GEMMS])
def test_production_gemm_shapes_take_the_production_kernel_and_match_fp32(ops, name, M, N, K, epi, has_bias, expect):
    """tolerance: 1.2e-2 x max(1, |ref|max) (bf16 output rounding 2^-9 rel. + accumulation order), as tests/test_gpu_ops.py"""
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(M * 31 + N)
    X = (torch.randn(M, K, generator=g, device=dev) * 0.7).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    b = (0.1 * torch.randn(N, generator=g, device=dev)).to(torch.bfloat16) if has_bias else None
    NO = N // 2 if epi == 'swiglu' else N
    R = torch.randn(M, NO, generator=g, device=dev).to(torch.bfloat16) if epi == 'resid' else None
    Xf, Wf = X.float(), W.float()
    if epi == 'swiglu':
        gate, up = W[:N // 2], W[N // 2:]
        Wi = torch.stack([gate.view(-1, 16, K), up.view(-1, 16, K)], 1).reshape(N, K).contiguous()      # the interleave mmd_finalize_weights builds
        Y = ops.gemm(X, Wi, None, epi=epi, variant=0)
        gg = (Xf @ gate.float().T).to(torch.bfloat16).float(); uu = (Xf @ up.float().T).to(torch.bfloat16).float()
        ref = F.silu(gg).to(torch.bfloat16).float() * uu
    else:
        Y = ops.gemm(X, W, b, R=R, epi=epi, variant=0)
        lin = F.linear(Xf, Wf, b.float() if b is not None else None)
        if epi == 'resid':
            ref = lin.to(torch.bfloat16).float() + R.float()
        elif epi == 'gelu_tanh':
            ref = O.gelu_tanh(lin.to(torch.bfloat16).float())
        elif epi == 'gelu_erf':
            ref = O.gelu_erf(lin.to(torch.bfloat16).float())
        else:
            ref = lin
    plan = _plan(ops)
    err = _rel_err(Y, ref)
    _record(f'gemm_{name}', M=M, N=N, K=K, epi=epi, rel_err=err, **plan)
    assert torch.isfinite(Y.float()).all()
    assert err <= 1.2e-2 * (2.0 if epi != 'none' else 1.0), (name, err, plan)
    if 'tiles' in expect:
        assert plan['kernel'] == 4 and plan['tiles'] == expect['tiles'], plan                                                    # GEMM_K_BIG64 in its 160-row form
    if 'min_tiles' in expect:
        assert plan['kernel'] in RING and plan['tiles'] >= expect['min_tiles'] and plan['blocks'] < plan['tiles'], plan      # persistent multi-tile loop ran
    if 'min_splits' in expect:
        assert plan['kernel'] in RING and plan['splits'] >= expect['min_splits'], plan                                        # automatic split-K over grid.z


def test_down_proj_of_merged_chunks_takes_the_split_ring():
    """Several streams' chunks in one forward (mmduet_amd/multistream.py): down_proj at M = 2548 is 140 output tiles -- over half a block wave, under the plain ring's threshold.
    The dispatcher splits K three ways there (420 work items in two rounds of K / 3; the 128-row kernel it fell back to ran at 0.28 of peak), given the large slab workspace
    of a model built for merged chunks.  Against fp32 math; M = 2352 (8 streams x 6 frames) likewise."""
    from rawops import RawOps
    ops_big = RawOps(torch.bfloat16, max_step_tokens=4096)
    dev = ops_big.dev
    for M in (2548, 2352):
        N, K = 3584, 18944
        g = torch.Generator(device=dev).manual_seed(M)
        X = (torch.randn(M, K, generator=g, device=dev) * 0.7).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
        Y = ops_big.gemm(X, W, None, R=R, epi='resid', variant=0)
        plan = _plan(ops_big)
        ref = (X.float() @ W.float().T).to(torch.bfloat16).float() + R.float()
        err = _rel_err(Y, ref)
        _record(f'gemm_llm_down_merged_M{M}', M=M, N=N, K=K, rel_err=err, **plan)
        assert plan['kernel'] in RING and plan['splits'] == 3, plan
        assert torch.isfinite(Y.float()).all() and err <= 2.4e-2, (M, err)
    del ops_big
    torch.cuda.empty_cache()


@pytest.mark.parametrize('variant', [32, 33])
@pytest.mark.parametrize('name,M,N,K,epi', [('vit_fc1', 25515, 4352, 1152, 'gelu_tanh'), ('vit_o', 25515, 1152, 1152, 'resid'), ('gate_up_tail', 1303, 37888, 3584, 'swiglu'),
                                            ('ragged', 3000, 1184, 704, 'none')])
def test_every_ring_instantiation_at_production_shapes(ops, variant, name, M, N, K, epi):
    """The two shipped gemm_ringx_kernel instantiations, forced (variant = GEMM_RINGX + flags: 16 = 8-wave 256x256, 17 = 4-wave 256x128), at production shapes:
    persistent multi-tile loops, M and N tails; both agree with fp32 math to the bf16 bound.  (The instantiations that lost their A/B -- late refill, 32x32x16 MFMA,
    four slots -- are no longer in the library: tools/probes/dropped/.)"""
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(N + K)
    X = (torch.randn(M, K, generator=g, device=dev) * 0.7).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    b = (0.1 * torch.randn(N, generator=g, device=dev)).to(torch.bfloat16)
    Xf = X.float()
    if epi == 'swiglu':
        gate, up = W[:N // 2], W[N // 2:]
        Wi = torch.stack([gate.view(-1, 16, K), up.view(-1, 16, K)], 1).reshape(N, K).contiguous()
        Y = ops.gemm(X, Wi, None, epi=epi, variant=variant)
        ref = F.silu((Xf @ gate.float().T).to(torch.bfloat16).float()).to(torch.bfloat16).float() * (Xf @ up.float().T).to(torch.bfloat16).float()
    elif epi == 'resid':
        R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
        Y = ops.gemm(X, W, b, R=R, epi=epi, variant=variant)
        ref = F.linear(Xf, W.float(), b.float()).to(torch.bfloat16).float() + R.float()
    else:
        Y = ops.gemm(X, W, b, epi=epi, variant=variant)
        lin = F.linear(Xf, W.float(), b.float())
        ref = O.gelu_tanh(lin.to(torch.bfloat16).float()) if epi == 'gelu_tanh' else lin
    plan = _plan(ops)
    err = _rel_err(Y, ref)
    assert torch.isfinite(Y.float()).all() and err <= 2.4e-2, (variant, name, err, plan)
    assert plan['kernel'] == (7 if variant & 1 else 6), plan


def test_ring_gemm_is_deterministic_and_tile_order_independent(ops):
    """Same operands twice -> identical bits (no atomics, fixed reduction order); forced single-tile-per-block launch (variant 6 at a shape
    with <= 256 tiles) vs the persistent loop on a sub-problem -> identical bits for the shared rows."""
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(5)
    X = torch.randn(25515, 1152, generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn(3456, 1152, generator=g, device=dev) / 34).to(torch.bfloat16)
    Y1 = ops.gemm(X, W, variant=0); p1 = _plan(ops)
    Y2 = ops.gemm(X, W, variant=0)
    assert torch.equal(Y1, Y2)
    Ys = ops.gemm(X[:2048], W, variant=6); p2 = _plan(ops)          # 8 x 14 = 112 tiles: one tile per block, no persistence
    assert p1['blocks'] < p1['tiles'] and p2['blocks'] == p2['tiles'] * p2['splits']
    assert torch.equal(Ys, Y1[:2048])


# ---- (b) chunk attention at the production sizes -----------------------------------------------------------------------------------
def _ref_attention_gpu(q, K, V, nh, nkv, d, n_ctx):
    """fp32 torch attention on the device, one kv group at a time.  q [S, nh*d]; K/V [nkv, cap, d]."""
    S = q.shape[0]; n_tot = n_ctx + S; rep = nh // nkv
    qh = q.float().view(S, nh, d).transpose(0, 1)
    out = torch.empty(nh, S, d, device=q.device)
    mask = torch.arange(n_tot, device=q.device)[None, :] > (torch.arange(S, device=q.device)[:, None] + n_ctx)
    for h in range(nkv):
        kk, vv = K[h, :n_tot].float(), V[h, :n_tot].float()
        s = qh[h * rep:(h + 1) * rep] @ kk.T * d ** -0.5
        s = s.masked_fill(mask[None], float('-inf'))
        out[h * rep:(h + 1) * rep] = torch.softmax(s, -1) @ vv
    return out.transpose(0, 1).reshape(S, nh * d)


# (147 rows x 7 heads = 1029 rows per kv head: the smallest step on the 256-row phase-split kernel; 146 stays on the 128-row form; 1 / 2 / 3 key tiles per split, a partial
#  last query block, new positions on tile boundaries, contexts deep enough for every ring slot to be refilled many times)
@pytest.mark.parametrize('S,n_ctx', [(1274, 0), (1274, 15000), (1274, 30000), (1323, 8000), (49, 30000), (131, 15000),
                                     (147, 0), (146, 0), (147, 1), (150, 63), (183, 64), (200, 100), (300, 70000), (637, 3), (1911, 27000), (2058, 127), (512, 4097)])
def test_chunk_attention_at_production_sizes(ops, S, n_ctx):
    """attn_gqa128_kernel (variant 3: the arena layout with V transposed in 64-token blocks) incl. the cost-model split-KV + merge.
    tolerance: 1.8e-2 x max(1, |ref|max): bf16 P and bf16 output rounding."""
    nh, nkv, d = 28, 4, 128
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    K[:, n_ctx + S:] = 1e4; V[:, n_ctx + S:] = 1e4            # beyond the valid range: must never reach the result
    o = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 3)
    ref = _ref_attention_gpu(q, K, V, nh, nkv, d, n_ctx)
    err = _rel_err(o, ref)
    _record(f'attn_S{S}_n{n_ctx}', rel_err=err)
    assert torch.isfinite(o.float()).all() and err <= 1.8e-2, err


@pytest.mark.parametrize('M', [49, 98, 196])
def test_stream_gemm_repeats_bit_identical_beside_a_copy_stream(ops, M):
    """Race screen of gemm_stream_kernel: the default for every 32 < M <= 256 GEMM relies on a hand-counted `s_waitcnt vmcnt` over inline-asm weight loads hipcc
    cannot see; a mis-count would show as a rare wrong tile, not as a parity failure.  Every shipped instantiation (M = 49 / 98 / 196 pick the 4- / 8- / 16-row-tile forms) in
    slab mode with a K split (qkv, o, down), with the SwiGLU epilogue (gate_up) and with the in-place epilogue (residual), 40 times beside a copy stream: the same bits."""
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(M)
    noise = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    cases = [('slabs', 4608, 3584), ('slabs', 3584, 3584), ('slabs', 3584, 18944), ('swiglu', 37888, 3584), ('resid', 3584, 18944)]
    for kind, N, K in cases:
        X = (torch.randn(M, K, generator=g, device=dev) * 0.7).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        R = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if kind == 'resid' else None

        def once():
            if kind == 'slabs':
                y, n = ops.gemm_slabs(X, W, variant=8)
                assert n >= 2                                   # a real K split
                return y
            if kind == 'swiglu':
                return ops.gemm(X, W, None, epi='swiglu', variant=8)
            return ops.gemm(X, W, None, R=R, epi='resid', variant=8)
        first = once().clone()
        assert _plan(ops)['kernel'] == 8                        # GEMM_K_STREAM
        for r in range(40):
            if r % 3 == 0:
                with torch.cuda.stream(side):
                    noise[:128 << 20].copy_(noise[128 << 20:], non_blocking=True)
            assert torch.equal(once(), first), (kind, N, K, r)
    torch.cuda.synchronize()


@pytest.mark.parametrize('name,M,N1,K1,epi1,N2,resid', [('vit_mlp', 25515, 4352, 1152, 'gelu_tanh', 1152, True), ('vit_mlp_last_layer_rows_NOT_on_the_ring', 6860, 4352, 1152, 'gelu_tanh', 1152, False),
                                                       ('llm_mlp', 1274, 37888, 3584, 'swiglu', 3584, True), ('llm_mlp_tail', 1303, 37888, 3584, 'swiglu', 3584, True),
                                                       ('llm_mlp_short_chunk', 700, 37888, 3584, 'swiglu', 3584, True)])
def test_piece_major_intermediate_gives_the_same_bits(ops, name, M, N1, K1, epi1, N2, resid):
    """The MLP pairs of the path (SigLIP fc1 + GELU -> fc2: models/live_llava/video_head_live_llava_qwen.py:96-98; Qwen2MLP gate_up + SwiGLU -> down) with the intermediate in the
    ring kernel's piece-major layout (1 KB pieces of 16 rows x 32 columns: the producer's wave instruction writes ONE contiguous piece, the consumer's X-DMA reads it back as one)
    against the row-major intermediate: the same values travel through other addresses -> the same bits; M tails (row groups past M are never written), both epilogues, the
    split-K consumer.  And against fp32 math."""
    import ctypes as C
    from mmduet_amd._lib import lib, check, EPI
    from mmduet_amd.modeling_live import _ptr
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(N1 + M)
    X = (torch.randn(M, K1, generator=g, device=dev) * 0.7).to(torch.bfloat16)
    W1 = (torch.randn(N1, K1, generator=g, device=dev) / math.sqrt(K1)).to(torch.bfloat16)
    T = N1 // 2 if epi1 == 'swiglu' else N1
    W2 = (torch.randn(N2, T, generator=g, device=dev) / math.sqrt(T)).to(torch.bfloat16)
    b1 = (0.1 * torch.randn(N1, generator=g, device=dev)).to(torch.bfloat16) if epi1 != 'swiglu' else None
    R = torch.randn(M, N2, generator=g, device=dev).to(torch.bfloat16) if resid else None
    W1k = W1
    if epi1 == 'swiglu':
        gate, up = W1[:T], W1[T:]
        W1k = torch.stack([gate.view(-1, 16, K1), up.view(-1, 16, K1)], 1).reshape(N1, K1).contiguous()
    outs = []
    for pm in (1, 0):
        Y = torch.empty(M, N2, device=dev, dtype=torch.bfloat16)
        used = C.c_int(-1)
        ops.m._bind_stream()
        check(lib().mmd_op_gemm_pair(ops.ctx, _ptr(X), _ptr(W1k), _ptr(b1), EPI[epi1], _ptr(W2), _ptr(R), _ptr(Y), M, N1, K1, N2, pm, C.byref(used)), ops.ctx, 'gemm_pair')
        torch.cuda.synchronize()
        # both GEMMs of the production pairs take the ring kernel; the 196-row last tower layer's fc2 (135 tiles: under one block wave) does not -- its intermediate stays row-major
        assert used.value == (pm if 'NOT_on_the_ring' not in name else 0), (name, pm, used.value)
        outs.append(Y)
    assert torch.equal(outs[0], outs[1])
    Xf = X.float()
    if epi1 == 'swiglu':
        mid = F.silu((Xf @ W1[:T].float().T).to(torch.bfloat16).float()).to(torch.bfloat16).float() * (Xf @ W1[T:].float().T).to(torch.bfloat16).float()
    else:
        mid = O.gelu_tanh(F.linear(Xf, W1.float(), b1.float()).to(torch.bfloat16).float())
    ref = mid.to(torch.bfloat16).float() @ W2.float().T
    if resid:
        ref = ref.to(torch.bfloat16).float() + R.float()
    err = _rel_err(outs[0], ref)
    assert torch.isfinite(outs[0].float()).all() and err <= 2.4e-2, (name, err)


def test_piece_major_is_refused_for_a_split_k_producer(ops):
    """a producer GEMM on the split-K ring (long K, few 256x256 tiles) leaves fp32 slabs and splitk_reduce writes its output ROW-major -- the pair must not
    report a piece-major intermediate there (no model pair has such a producer: fc1 has K = 1152, gate_up is SwiGLU; the exported mmd_op_gemm_pair reaches it)."""
    import ctypes as C
    from mmduet_amd._lib import lib, check, EPI
    from mmduet_amd.modeling_live import _ptr
    dev = ops.dev
    M, N1, K1, N2 = 1024, 2048, 8192, 12288
    g = torch.Generator(device=dev).manual_seed(77)
    X = (torch.randn(M, K1, generator=g, device=dev) * 0.7).to(torch.bfloat16)
    W1 = (torch.randn(N1, K1, generator=g, device=dev) / math.sqrt(K1)).to(torch.bfloat16)
    W2 = (torch.randn(N2, N1, generator=g, device=dev) / math.sqrt(N1)).to(torch.bfloat16)
    outs = []
    for pm in (1, 0):
        Y = torch.empty(M, N2, device=dev, dtype=torch.bfloat16)
        used = C.c_int(-1)
        ops.m._bind_stream()
        check(lib().mmd_op_gemm_pair(ops.ctx, _ptr(X), _ptr(W1), None, EPI['none'], _ptr(W2), None, _ptr(Y), M, N1, K1, N2, pm, C.byref(used)), ops.ctx, 'gemm_pair')
        torch.cuda.synchronize()
        assert used.value == 0, (pm, used.value)
        outs.append(Y)
    assert torch.equal(outs[0], outs[1])
    ref = (X.float() @ W1.float().T).to(torch.bfloat16).float() @ W2.float().T
    assert _rel_err(outs[0], ref) <= 2.4e-2


# (per-frame steps and short chunks: attn_gqa128_w1_kernel, variant 5 = forced.  1 / 2 / 3 row blocks, blocks whose waves hold 2 / 1 / 0 row tiles, 1 .. 8 key tiles per split,
#  new positions on tile boundaries, the masked diagonal inside the first / second half tile, a context too short for any split)
W1_SHAPES = [(49, 15000), (49, 0), (49, 1), (49, 63), (49, 64), (49, 4047), (49, 30000), (24, 15000), (64, 4096), (98, 15000), (109, 8000), (131, 15000), (17, 300), (3, 70001),
             (37, 129), (256, 5000), (300, 70000), (1274, 0)]


@pytest.mark.parametrize('S,n_ctx', W1_SHAPES)
def test_w1_attention_at_production_sizes(ops, S, n_ctx):
    """attn_gqa128_w1_kernel (attn_w1.h: software-pipelined waves, balanced row blocks; Qwen2Attention.forward, transformers qwen2/modeling_qwen2.py:200-240) against fp32
    math, same bound as the other forms.  (Inside the model the dispatch takes it for 17 .. 768 stacked rows over >= 4096 keys: tests/test_gpu_fullsize.py runs those steps.)"""
    nh, nkv, d = 28, 4, 128
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    K[:, n_ctx + S:] = 1e4; V[:, n_ctx + S:] = 1e4            # beyond the valid range: must never reach the result
    o = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 5)
    ref = _ref_attention_gpu(q, K, V, nh, nkv, d, n_ctx)
    err = _rel_err(o, ref)
    _record(f'attn_w1_S{S}_n{n_ctx}', rel_err=err)
    assert torch.isfinite(o.float()).all() and err <= 1.8e-2, err
    # ... and it agrees with the phase-split / two-slot forms (variant 3) to accumulation-order noise: same fragments, same rounding points, other unit order
    old = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 3)
    assert _rel_err(o, old.float()) <= 8e-3


def test_model_steps_take_the_attention_form_meant_for_them(width2):
    """Inside the model (true widths, arena layout, the dispatcher's own choice): a per-frame step over a long context runs attn_gqa128_w1_kernel with two balanced row
    blocks (343 stacked rows per kv head), a 26-frame chunk the 256-row phase-split form in its contiguous decomposition (attn_gqa128_chunk_kernel), a decode row the loader / compute ring, a step over a short context the two-slot
    form -- mmd_op_attention_last_form is to the attention what mmd_op_gemm_last_plan is to the GEMMs."""
    import ctypes as C
    from mmduet_amd._lib import lib, check
    m = width2[0]
    A = m.new_cache(initial_tokens=15000 + 4096)
    check(lib().mmd_kv_debug_set_len(A.arena.h, 15000), m._ctx, 'set_len')          # (the slots hold zeros: same traffic, same kernels)
    form = (C.c_int * 2)()
    for S, want in ((49, 5), (98, 5), (109, 5), (110, 4), (1274, 8), (147, 4), (1, 3), (2, 3)):
        x = (torch.randn(1, S, m.config.hidden_size, device=m.device) * 0.5).to(torch.bfloat16)
        out = m(inputs_embeds=x, past_key_values=type(A)(A.arena, 15000))
        torch.cuda.synchronize()
        assert torch.isfinite(out.informative_logits).all()
        lib().mmd_op_attention_last_form(m._ctx, form)
        assert form[0] == want, (S, list(form))
        if want == 5:
            assert form[1] >= 8                      # key splits: two row blocks x 4 kv heads fill the chip
    B = m.new_cache(initial_tokens=8192)
    check(lib().mmd_kv_debug_set_len(B.arena.h, 1000), m._ctx, 'set_len')
    m(inputs_embeds=(torch.randn(1, 49, m.config.hidden_size, device=m.device) * 0.5).to(torch.bfloat16), past_key_values=type(B)(B.arena, 1000))
    torch.cuda.synchronize(); lib().mmd_op_attention_last_form(m._ctx, form)
    assert form[0] == 4                              # under 4096 keys the two-slot form stays (shorter pipeline fill)


# (multi-frame chunks in the contiguous decomposition of attn_chunk.h, variant 6 = forced: units cut in 2 - 3 parts, units that one block covers alone (written straight to the
#  output), many short units per block (first chunk of a stream), a single kv-head row block, ragged last row block, contexts on / off tile boundaries)
CHUNK_SHAPES = [(1274, 0), (1274, 15000), (1274, 30000), (1323, 8000), (147, 0), (147, 1), (150, 63), (183, 64), (200, 100), (300, 70000), (637, 3), (1911, 27000), (2058, 127),
                (512, 4097), (37, 129), (392, 15000), (40, 5000)]


@pytest.mark.parametrize('S,n_ctx', CHUNK_SHAPES)
def test_chunk_kernel_contiguous_decomposition(ops, S, n_ctx):
    """attn_gqa128_chunk_kernel + attn_combine128_chunk_kernel against fp32 math (same bound as the grid form) and against the grid form itself (accumulation-order noise)."""
    nh, nkv, d = 28, 4, 128
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    K[:, n_ctx + S:] = 1e4; V[:, n_ctx + S:] = 1e4            # beyond the valid range: must never reach the result
    o = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 6)
    ref = _ref_attention_gpu(q, K, V, nh, nkv, d, n_ctx)
    err = _rel_err(o, ref)
    _record(f'attn_chunk_S{S}_n{n_ctx}', rel_err=err)
    assert torch.isfinite(o.float()).all() and err <= 1.8e-2, err
    assert _rel_err(o, ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 3).float()) <= 8e-3


@pytest.mark.parametrize('S,n_ctx', [(1274, 15000), (1274, 1100), (637, 9000)])
def test_chunk_kernel_repeats_bit_identical_beside_a_copy_stream(ops, S, n_ctx):
    """Race screen of attn_gqa128_chunk_kernel: the ring restarts at every segment of a block's range (a barrier between the last P.V reads and the next stages)."""
    nh, nkv, d = 28, 4, 128
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    noise = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    first = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 6).clone()
    for r in range(60):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                noise[:128 << 20].copy_(noise[128 << 20:], non_blocking=True)
        assert torch.equal(ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 6), first), r
    torch.cuda.synchronize()


@pytest.mark.parametrize('S,n_ctx', [(49, 15000), (98, 30000), (24, 4100)])
def test_w1_attention_repeats_bit_identical_beside_a_copy_stream(ops, S, n_ctx):
    """Race screen of attn_gqa128_w1_kernel's LDS-DMA ring (hand-counted vmcnt, one barrier per tile, fragments read across phase boundaries): see the ring forms' screen below."""
    nh, nkv, d = 28, 4, 128
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    noise = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    first = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 5).clone()
    for r in range(60):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                noise[:128 << 20].copy_(noise[128 << 20:], non_blocking=True)
        assert torch.equal(ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 5), first), r
    torch.cuda.synchronize()


@pytest.mark.parametrize('S,n_ctx', [(1274, 15000), (637, 3000), (1, 15000), (2, 70001)])
def test_ring_attention_repeats_bit_identical_beside_a_copy_stream(ops, S, n_ctx):
    """Race screen of the LDS-DMA rings (chunk kernel <2, 4, 8>, decode kernel <1, 4>): counted waits and slot reuse are hand-placed, and a DMA landing late or a slot refilled
    early would show as a rare differing tile, not as a parity failure -- so the same launch is repeated with a copy stream perturbing the memory system and must give the
    same BITS every time (tools/probes/attn_race_screen.py runs the long version: 8 shapes x 300 repeats)."""
    nh, nkv, d = 28, 4, 128
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    noise = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    first = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 3).clone()
    for r in range(60):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                noise[:128 << 20].copy_(noise[128 << 20:], non_blocking=True)
        assert torch.equal(ops.attention(q, K, V, nh, nkv, d, n_ctx, True, 3), first), r
    torch.cuda.synchronize()


def _ref_attention_rows(q, K, V, nh, nkv, d, n_ctx, rows):
    """fp32 attention of the query rows `rows` only (row r sees keys 0 .. n_ctx + r), one kv group and 256 Ki keys at a time with a running (max, sum) --
    the 1 M-key contexts never materialise an [S, n] score matrix.  q [S, nh*d]; K / V [nkv, cap, d] row-major."""
    rep = nh // nkv
    rows_t = torch.as_tensor(rows, device=q.device)
    qh = q.float().view(q.shape[0], nh, d)[rows_t].transpose(0, 1)              # [nh, R, d]
    out = torch.empty(nh, len(rows), d, device=q.device)
    n_max = n_ctx + max(rows) + 1
    for h in range(nkv):
        qq = qh[h * rep:(h + 1) * rep] * d ** -0.5
        m = torch.full((rep, len(rows), 1), float('-inf'), device=q.device); l = torch.zeros_like(m); acc = torch.zeros(rep, len(rows), d, device=q.device)
        for k0 in range(0, n_max, 1 << 18):
            k1 = min(n_max, k0 + (1 << 18))
            s = qq @ K[h, k0:k1].float().T
            dead = torch.arange(k0, k1, device=q.device)[None, :] > (rows_t[:, None] + n_ctx)
            s = s.masked_fill(dead[None], float('-inf'))
            m2 = torch.maximum(m, s.amax(-1, keepdim=True))
            p = torch.exp(s - m2); sc = torch.exp(m - m2)
            l = l * sc + p.sum(-1, keepdim=True); acc = acc * sc + p @ V[h, k0:k1].float(); m = m2
        out[h * rep:(h + 1) * rep] = acc / l
    return out.transpose(0, 1).reshape(len(rows), nh * d)


@pytest.mark.parametrize('S,variant,form', [(49, 5, 5), (1, 3, 3), (1274, 6, 8)], ids=['frame_step', 'decode_row', 'chunk_26_frames'])
def test_attention_over_a_million_keys(ops, S, variant, form):
    """BASELINE configs[2] names "KV-cache growth to HBM limit" (test/inference.py:239 over the growing cache): the three LLM attention forms of the stream -- a 49-row frame
    step (attn_gqa128_w1_kernel), a decode row (the loader / compute ring) and a 26-frame chunk (attn_gqa128_chunk_kernel) -- over 1 000 037 keys (57 GB of a 7B stream's
    arena, 20 000 frames) against fp32 math on sampled query rows; same bound as at the production sizes.  Also: unused slots behind the context never reach the result."""
    import ctypes as C
    from mmduet_amd._lib import lib
    nh, nkv, d = 28, 4, 128
    n_ctx = 1_000_037
    dev = ops.dev
    g = torch.Generator(device=dev).manual_seed(S + 7)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    K[:, n_ctx + S:] = 1e4; V[:, n_ctx + S:] = 1e4
    # make a handful of keys spread over the whole context matter (random scores over 1 M keys are nearly flat): every sampled row gets keys it strongly prefers
    rows = sorted({0, S - 1, S // 2, S // 3, (2 * S) // 3})
    hot = torch.randint(0, n_ctx, (24,), generator=g, device=dev)
    qf = q.float().view(S, nh, d)
    for j, r in enumerate(rows):
        for h in range(nkv):
            K[h, hot[(j * nkv + h) % 24]] = (qf[r, h * (nh // nkv)] * 1.5).to(torch.bfloat16)          # score ~ 17 against ln(sum of a million e^N(0,1)) ~ 14.3
    o = ops.attention(q, K, V, nh, nkv, d, n_ctx, True, variant)
    got_form = (C.c_int * 2)(); lib().mmd_op_attention_last_form(ops.ctx, got_form)
    assert got_form[0] == form, list(got_form)
    ref = _ref_attention_rows(q, K, V, nh, nkv, d, n_ctx, rows)
    err = _rel_err(o[torch.as_tensor(rows, device=dev)], ref)
    _record(f'attn_1M_S{S}', rel_err=err, form=got_form[0], splits=got_form[1])
    assert torch.isfinite(o.float()).all() and err <= 1.8e-2, err
    assert ref.abs().max().item() > 0.05          # the sampled rows are not the flat average of a million values


# ---- true-width models ---------------------------------------------------------------------------------------------------------------
def _build(llm_layers, vit_layers, dtype, vocab=2048, max_vit_batch=35, max_step_tokens=1536, seed=3, tower_dtype=None, weight_dtype=None):
    """(HIP model, oracle weights dict on the device in `dtype`-rounded fp32, oracle config)."""
    from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
    from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
    from mmduet_amd.weights import synthetic_weights
    pcfg = VideoHeadLiveLlavaQwenConfig(vocab_size=vocab, num_hidden_layers=llm_layers, vit_num_hidden_layers=vit_layers + 1, vit_layers_removed=1,
                                        frame_num_tokens=49, frame_resolution=384, v_placeholder='<image>')
    ocfg = O.OracleConfig(vocab_size=vocab, num_hidden_layers=llm_layers, vit_layers=vit_layers)
    if tower_dtype:
        pcfg.tower_dtype = tower_dtype
    if weight_dtype:
        pcfg.weight_dtype = weight_dtype
    m = VideoHeadLiveLlavaQwenForCausalLM(pcfg, torch_dtype=dtype, max_vit_batch=max_vit_batch, max_step_tokens=max_step_tokens, kv_initial_tokens=4096)
    w = {}
    for name, t in synthetic_weights(pcfg, seed=seed, device=m.device, dtype=dtype, scale='unit'):
        m.load_tensor(name, t)
        w[name] = t
    m.finalize()
    return m, w, ocfg


def _oracle(w, ocfg, dtype):
    return O.OracleModel(ocfg, {k: v.to(dtype) for k, v in w.items()})


@pytest.fixture(scope='module')
def width2():
    """true widths, 2 tower layers + 2 decoder layers, bf16"""
    m, w, ocfg = _build(2, 2, torch.bfloat16)
    yield m, _oracle(w, ocfg, torch.float32), _oracle(w, ocfg, torch.bfloat16)
    del m, w
    torch.cuda.empty_cache()


def maxerr(a, b):
    return (a.float() - b.float().to(a.device)).abs().max().item()


def test_tower_batch_of_35_frames_true_width(width2):
    """(c) mmd_vit_encode on the production batch (M = 25 515 rows: persistent ring GEMMs, batch-35 ViT attention, projector, pooling) vs the
    oracle in fp32 and in bf16.  tolerance: |ours - fp32 oracle| <= 3 x |bf16 oracle - fp32 oracle| + 2e-2 x scale."""
    m, o32, o16 = width2
    g = torch.Generator(device=m.device).manual_seed(0)
    px = torch.randn(35, 3, 384, 384, generator=g, device=m.device).to(torch.bfloat16)
    ve = m.visual_embed(px)
    r32 = o32.visual_embed(px.float())
    r16 = o16.visual_embed(px)
    scale = r32.abs().max().item()
    e_ours, e_ref = maxerr(ve, r32), maxerr(r16, r32)
    _record('tower_35_frames_2_layers', ours_vs_fp32=e_ours, bf16_oracle_vs_fp32=e_ref, scale=scale)
    assert ve.shape == (35 * 49, 3584)
    assert e_ours <= 3 * e_ref + 2e-2 * max(1.0, scale), (e_ours, e_ref, scale)


def test_vit_ring_attention_repeats_bit_identical_beside_a_copy_stream(width2):
    """Race screen of attn_d72_ring_kernel (two-slot LDS-DMA ring, one hand-placed wait + raw barrier per tile, V fragments read from an asm block the compiler cannot see
    into): the 35-frame tower (2 layers: 2 x 3 360 blocks per pass, incl. the 196-row last layer) and a 3-frame one are repeated beside a copy stream that perturbs the
    memory system; a DMA landing late or a slot refilled early would show as a rare differing tile, not as a parity failure."""
    m, _, _ = width2
    dev = m.device
    g = torch.Generator(device=dev).manual_seed(9)
    noise = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    for nf, reps in ((35, 40), (3, 60)):
        px = torch.randn(nf, 3, 384, 384, generator=g, device=dev).to(torch.bfloat16)
        for full in (False, True):
            m.set_full_tower(full)
            first = m.visual_embed(px).clone()
            for r in range(reps):
                if r % 3 == 0:
                    with torch.cuda.stream(side):
                        noise[:128 << 20].copy_(noise[128 << 20:], non_blocking=True)
                assert torch.equal(m.visual_embed(px), first), (nf, full, r)
    m.set_full_tower(False)
    torch.cuda.synchronize()


def test_vit_ring_attention_equals_register_staged_kernel():
    """attn_d72_ring_kernel (K / V by LDS-DMA, row-major V image, 16-deep MFMA for dims 64..71) against attn_rowmajor_kernel<3, 5> (MMDUET_VIT_ATTN_RING=0) inside the
    tower at the true widths: fp16 and bf16 tower, 35 frames (6 query blocks x 16 heads x 35, last key tile 25 of 64) and 3 frames, the pooled output of the sparse last
    layer (196 query rows over 729 keys) and of the full tower.  Same products; since round 6 the ring kernel adds the dims 64..71 part of a score with one fp32 v_add behind the
    64-dim chain (D72_TAIL_SEPARATE: no MFMA reads a different-depth MFMA's result) where the register-staged kernel accumulates it inside the chain -- the same fp32 terms in
    another order, so the outputs agree to fp32-rounding noise carried through two layers (<= 2 bf16 ulps on a few elements), no longer bit for bit."""
    import subprocess, sys, hashlib
    code = r'''
import os, sys, json, hashlib, torch
sys.path.insert(0, os.environ["MMD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MMD_ROOT"], "tests"))
from test_gpu_production import _build
res = {}
for tower in ("fp16", "bf16"):
    m, w, ocfg = _build(1, 2, torch.bfloat16, max_vit_batch=35, max_step_tokens=64, tower_dtype=tower)
    g = torch.Generator(device=m.device).manual_seed(5)
    for nf in (35, 3):
        px = torch.randn(nf, 3, 384, 384, generator=g, device=m.device).to(torch.bfloat16)
        for full in (False, True):
            m.set_full_tower(full)
            y = m.visual_embed(px); torch.cuda.synchronize()
            assert torch.isfinite(y.float()).all()
            res[f"{tower}_{nf}_{int(full)}"] = y.float().cpu().flatten()[::7].tolist()
    m.set_full_tower(False)
    del m, w; torch.cuda.empty_cache()
print("RES " + json.dumps(res))
'''
    def run(**kw):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MMD_ROOT=ROOT, **kw), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith('RES ')][0][4:])
    ring, staged = run(), run(MMDUET_VIT_ATTN_RING='0')
    assert len(ring) == 8
    for k in ring:
        a, b = torch.tensor(ring[k]), torch.tensor(staged[k])
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 2 ** -6 * scale, (k, (a - b).abs().max().item(), scale)           # <= 2 ulps of the bf16 output at its largest magnitude
        assert (a - b).abs().mean().item() <= 2 ** -10 * scale, (k, (a - b).abs().mean().item())                # ... and far below one ulp on average (one-ulp flips on ~10 % of the
                                                                                                              # elements after two layers + projector): the kernels are the same arithmetic
    # and the ring kernel itself is deterministic
    again = run()
    assert all(again[k] == ring[k] for k in ring)


def test_chunk_of_26_frames_equals_26_frame_steps_true_width(width2):
    """(d) k = 26 (M = 1274 + prefix: ring GEMMs, gqa128 chunk attention) vs 26 one-frame steps (weight-streaming kernels) vs the fp32 oracle.
    tolerance on head logits: 6e-2 (bf16, 2 layers), and chunk == per-frame within the same bound."""
    m, o32, o16 = width2
    g = torch.Generator(device=m.device).manual_seed(2)
    frames = [(torch.randn(49, 3584, generator=g, device=m.device) * 0.5).to(torch.bfloat16) for _ in range(26)]
    prompt = (torch.randn(1, 29, 3584, generator=g, device=m.device) * 0.5).to(torch.bfloat16)
    base = m(inputs_embeds=prompt).past_key_values
    per, cache = [], base
    for f in frames:
        sc, cache = m.frame_step(f[None], cache, [48]); per.append(sc[0])
    per = torch.stack(per)
    rows = [49 * (j + 1) - 1 for j in range(26)]
    chunk, cache2 = m.frame_step(torch.cat(frames)[None], m.cache_prefix(base, len(base)), rows)
    assert len(cache2) == len(cache) == 29 + 26 * 49
    oc = o32(inputs_embeds=prompt.float()).past_key_values
    ref = o32(inputs_embeds=torch.cat(frames)[None].float(), past_key_values=oc)
    want = torch.cat([ref.informative_logits[0, rows], ref.relevance_logits[0, rows]], -1).cpu()
    ref16 = o16(inputs_embeds=torch.cat(frames)[None], past_key_values=o16(inputs_embeds=prompt).past_key_values)
    want16 = torch.cat([ref16.informative_logits[0, rows], ref16.relevance_logits[0, rows]], -1).cpu()
    _record('chunk26_2_layers', chunk_vs_fp32=maxerr(chunk, want), per_frame_vs_fp32=maxerr(per, want), chunk_vs_per_frame=maxerr(chunk, per),
            bf16_oracle_vs_fp32=maxerr(want16, want))
    assert maxerr(chunk, want) < 6e-2 and maxerr(per, want) < 6e-2 and maxerr(chunk, per) < 6e-2


@pytest.fixture(scope='module')
def width2_fp8():
    """true widths, 1 tower layer + 2 decoder layers, fp8-e4m3 decoder weights (BASELINE configs[4])"""
    m, w, ocfg = _build(2, 1, torch.bfloat16, weight_dtype='fp8_e4m3')
    yield (m,)
    del m, w
    torch.cuda.empty_cache()


@pytest.mark.parametrize('layout', ['four_staggered_with_chunks', 'six_together', 'four_staggered_with_chunks_fp8', 'six_together_fp8'])
def test_native_decode_rounds_equal_single_stream_generate_true_width(request, layout):
    width2 = request.getfixturevalue('width2_fp8' if layout.endswith('_fp8') else 'width2')
    layout = layout.replace('_fp8', '')
    """mmd_round_multi (several streams per forward, sampling on the device; the multi-stream form of models/modeling_live.py:51-77) against mmd_greedy_generate run
    stream by stream on the same contexts, at true widths in bf16.  The rounds cover the three schedules a round can take: every talking stream's row alone (<= 4 rows:
    the GEMV chain with the q / k / v preparation inside the attention kernel, each stream's rows of the qkv slabs at its row offset), talking rows next to a 98-row
    frame chunk (the fused slab schedule, <= 256 rows) and next to a 637-row chunk (tile GEMMs).  Token ids equal, the repetition-penalty list carried over a second
    response, KV lengths equal, and the arenas continue to the same head logits afterwards."""
    m = width2[0]
    H = m.config.hidden_size
    g = torch.Generator(device=m.device).manual_seed(11)
    rnd = lambda n: (torch.randn(n, H, generator=g, device=m.device) * 0.5).to(torch.bfloat16)
    ctx_x = [rnd(5000), rnd(700), rnd(300), rnd(1200)]              # the four streams' pasts (5000: the attention runs its long-context forms)
    prompts = [rnd(4), rnd(6), rnd(4), rnd(5)]
    starts = (0, 0, 2, 5)
    if layout == 'six_together':          # six talking streams from the first round on: 6 rows per round = the slab schedule (not the GEMV chain) with the batched decode attention
        ctx_x += [rnd(64), rnd(4100)]; prompts = [rnd(4) for _ in range(6)]; starts = (0,) * 6
    n_str = len(ctx_x)
    chunk98, chunk637, probe = rnd(98), rnd(637), rnd(49)
    N = 10
    import ctypes as C
    from mmduet_amd._lib import lib
    forms = set()

    def build():
        caches = []
        for x in ctx_x:
            c = None
            for s0 in range(0, x.shape[0], 1024):
                c = m(inputs_embeds=x[None, s0:s0 + 1024], past_key_values=c).past_key_values
            caches.append(c)
        return caches

    # reference: one stream at a time through the single-stream native loop (two responses each: the penalty list persists)
    ref_ids, ref_len, ref_probe, ref_seen = [], [], [], []
    caches = build()
    for c, p in zip(caches, prompts):
        seen = [3, 5]
        ids1, c1 = m.greedy_generate(p, c, -1, N, 1.15, seen)
        ids2, c2 = m.greedy_generate(p, c1, -1, N, 1.15, seen)
        ref_ids.append((ids1, ids2)); ref_len.append(len(c2)); ref_seen.append(list(seen))
        ref_probe.append(m.frame_step(probe[None], c2, [48])[0])
    # the same through rounds: streams 0 / 1 start together, stream 2 two rounds later, stream 3 five rounds later; a 98-row chunk of a watching stream rides in
    # rounds 3-4, a 637-row chunk in round 6
    caches = build()
    smp = [m.new_sampler() for _ in range(n_str)]
    watcher = m.new_cache()
    got = [([], []) for _ in range(n_str)]
    seen = [[3, 5] for _ in range(n_str)]
    for resp in range(2):
        state = [dict(start=st, n=0) for st in starts]
        rnd_i = 0
        while any(s['n'] < N for s in state):
            segs, who = [], []
            for i, s in enumerate(state):
                if rnd_i < s['start'] or s['n'] >= N:
                    continue
                if s['n'] == 0:
                    smp[i].begin(-1, 1.15, seen[i], N)
                    segs.append(dict(x=prompts[i], cache=caches[i], sampler=smp[i], sample=True))
                else:
                    segs.append(dict(x=None, cache=caches[i], sampler=smp[i], feed=True, sample=True))
                who.append(i)
            if rnd_i in (3, 4) and layout != 'six_together':
                segs.append(dict(x=chunk98, cache=watcher, head_rows=[48, 97]))
            if rnd_i == 6 and layout != 'six_together':
                segs.insert(1, dict(x=chunk637, cache=watcher, head_rows=[636])); who.insert(1, None)
            out = m.round_multi(segs)
            f = (C.c_int * 2)(); lib().mmd_op_attention_last_form(m._ctx, f); forms.add(f[0])
            for i, o in zip(who + [None] * (len(segs) - len(who)), out):
                if i is None:
                    if o['heads'] is not None:
                        watcher = o['cache']; assert torch.isfinite(o['heads']).all()
                    continue
                caches[i] = o['cache']; got[i][resp].append(o['token']); state[i]['n'] += 1
                seen[i].append(o['token'])
            rnd_i += 1
    assert 9 in forms                      # the talking streams' rows really shared one attention launch (launch_attention_decode_multi)
    emb = m.get_input_embeddings()

    def near_tie(i, resp, t, a, b):
        """bf16: the rounds sum the same products in another order (other K splits of the GEMVs, other key ranges per attention partial), so two candidates within rounding
        noise may swap.  Teacher-forced logits of the single-stream model at the point of divergence: the two tokens must be that close."""
        r0, r1 = ref_ids[i]
        rows = [ctx_x[i], prompts[i]]
        pen_list = [3, 5]
        if resp == 1:
            rows += [emb(torch.tensor(r0[:-1], device=m.device)).view(-1, H), prompts[i]]; pen_list += r0
        cur = (r0, r1)[resp]
        if t:
            rows.append(emb(torch.tensor(cur[:t], device=m.device)).view(-1, H)); pen_list += cur[:t]
        lg = m(inputs_embeds=torch.cat(rows)[None]).logits[0, -1].float()
        idx = torch.as_tensor(pen_list, device=lg.device)
        lg[idx] = torch.where(lg[idx] < 0, lg[idx] * 1.15, lg[idx] / 1.15)
        return abs(lg[a].item() - lg[b].item()) <= 0.02 * max(1.0, lg.abs().max().item())

    diverged = 0
    for i in range(n_str):
        ok = True
        for resp in range(2):
            if got[i][resp] != ref_ids[i][resp]:
                t = next(k for k, (x, y) in enumerate(zip(got[i][resp], ref_ids[i][resp])) if x != y)
                assert near_tie(i, resp, t, got[i][resp][t], ref_ids[i][resp][t]), (i, resp, t, got[i][resp], ref_ids[i][resp])
                ok = False; diverged += 1
                break
        assert len(caches[i]) == ref_len[i]
        if not ok:
            continue                       # (a swapped near-tie changes everything behind it)
        assert seen[i] == ref_seen[i]
        pr = m.frame_step(probe[None], caches[i], [48])[0]
        assert maxerr(pr, ref_probe[i]) <= 0.06 * max(1.0, ref_probe[i].abs().max().item()), i          # (bf16: the rounds' GEMVs ran over 1 .. 640 rows, other accumulation order)
    assert diverged <= 2          # (each one verified above as a near-tie of the single-stream logits; a plumbing error diverges everywhere and by a wide margin)
    assert len(watcher) == (0 if layout == 'six_together' else 2 * (98 * 2 + 637))


def test_batched_decode_attention_repeats_bit_identical_beside_a_copy_stream(width2):
    """Race screen of launch_attention_decode_multi (the decode ring kernel with grid.x = stream: per-stream context / arena from a device table, partials [stream][split][rows],
    each stream's q / k / v prepared from its rows of the shared qkv slabs, the new K / V rows appended by the block whose key range holds the position): the same four-stream
    decode round -- GEMV chain, 4 rows -- and the same six-stream round (slab schedule) from the same state, 40 times beside a copy stream: the same bits in every hidden row."""
    import ctypes as C
    from mmduet_amd._lib import lib
    m = width2[0]
    H = m.config.hidden_size
    g = torch.Generator(device=m.device).manual_seed(31)
    rnd = lambda n: (torch.randn(n, H, generator=g, device=m.device) * 0.5).to(torch.bfloat16)
    lens = [5000, 700, 64, 1200, 4100, 333]
    caches = []
    for n in lens:
        c = None
        x = rnd(n)
        for s0 in range(0, n, 1024):
            c = m(inputs_embeds=x[None, s0:s0 + 1024], past_key_values=c).past_key_values
        caches.append(c)
    rows = [rnd(1) for _ in lens]
    noise = torch.empty(256 << 20, dtype=torch.uint8, device=m.device)
    side = torch.cuda.Stream()
    form = (C.c_int * 2)()
    for n_str in (4, 6):
        first = None
        for r in range(40):
            if r % 3 == 0:
                with torch.cuda.stream(side):
                    noise[:128 << 20].copy_(noise[128 << 20:], non_blocking=True)
            out = m.multi_step([dict(x=rows[i], cache=m.cache_prefix(caches[i], lens[i]), hidden='all') for i in range(n_str)], want_logits=False)
            hid = torch.cat([o['hidden'] for o in out]).clone()
            lib().mmd_op_attention_last_form(m._ctx, form)
            assert form[0] == 9, list(form)
            assert torch.isfinite(hid.float()).all()
            if first is None:
                first = hid
            else:
                assert torch.equal(hid, first), (n_str, r)
    torch.cuda.synchronize()


def test_per_frame_steps_over_a_long_context_true_width(width2):
    """The reference's own schedule -- ONE frame per forward (test/inference.py:221-246) -- deep into a stream: three 1372-row chunks build a 4.1 k-token context, then eight
    49-row frame steps and a 98-row step run on attn_gqa128_w1_kernel (asserted), against the fp32 oracle fed the same inputs; the bf16 oracle (the reference's eager rounding
    points) is the yardstick.  Same bound as the chunk test above: 6e-2 on head logits (bf16, 2 layers)."""
    import ctypes as C
    from mmduet_amd._lib import lib
    m, o32, o16 = width2
    g = torch.Generator(device=m.device).manual_seed(9)
    rnd = lambda S: (torch.randn(1, S, 3584, generator=g, device=m.device) * 0.5).to(torch.bfloat16)
    cache = oc = oc16 = None
    for _ in range(3):
        x = rnd(1372)
        cache = m(inputs_embeds=x, past_key_values=cache).past_key_values
        oc = o32(inputs_embeds=x.float(), past_key_values=oc).past_key_values
        oc16 = o16(inputs_embeds=x, past_key_values=oc16).past_key_values
    form = (C.c_int * 2)()
    worst = worst16 = 0.0
    for S in [49] * 8 + [98]:
        x = rnd(S)
        out = m(inputs_embeds=x, past_key_values=cache); cache = out.past_key_values
        torch.cuda.synchronize(); lib().mmd_op_attention_last_form(m._ctx, form)
        assert form[0] == 5, (S, list(form))
        ref = o32(inputs_embeds=x.float(), past_key_values=oc); oc = ref.past_key_values
        r16 = o16(inputs_embeds=x, past_key_values=oc16); oc16 = r16.past_key_values
        for a, b, c16 in ((out.informative_logits, ref.informative_logits, r16.informative_logits), (out.relevance_logits, ref.relevance_logits, r16.relevance_logits)):
            worst = max(worst, maxerr(a[0, -1], b[0, -1])); worst16 = max(worst16, maxerr(c16[0, -1].float(), b[0, -1]))
    assert len(cache) == 3 * 1372 + 8 * 49 + 98
    _record('per_frame_long_context_2_layers', ours_vs_fp32=worst, bf16_oracle_vs_fp32=worst16)
    assert worst < 6e-2, (worst, worst16)


def test_fp32_mode_true_width_meets_1e3():
    """(f) fp32 build at the true widths (2 + 2 layers): head logits within 1e-3 of the fp32 oracle (the north-star tolerance), frame chunk and decode rows."""
    m, w, ocfg = _build(2, 2, torch.float32, max_vit_batch=2, max_step_tokens=512)
    o32 = _oracle(w, ocfg, torch.float32)
    g = torch.Generator(device=m.device).manual_seed(4)
    px = torch.randn(2, 3, 384, 384, generator=g, device=m.device)
    ve = m.visual_embed(px)
    rv = o32.visual_embed(px)
    e_ve = maxerr(ve, rv)
    cache = ocache = None
    worst = 0.0
    steps = [torch.randn(1, 31, 3584, generator=g, device=m.device) * 0.5, ve[:49][None], ve[49:][None], torch.randn(1, 1, 3584, generator=g, device=m.device) * 0.5]
    for x in steps:
        out = m(inputs_embeds=x, past_key_values=cache); cache = out.past_key_values
        ref = o32(inputs_embeds=x, past_key_values=ocache); ocache = ref.past_key_values
        worst = max(worst, maxerr(out.informative_logits[0, -1], ref.informative_logits[0, -1]), maxerr(out.relevance_logits[0, -1], ref.relevance_logits[0, -1]))
    _record('fp32_mode_true_width', visual_embed_err=e_ve, visual_embed_scale=rv.abs().max().item(), head_logit_err=worst)
    assert worst <= 1e-3, worst
    assert e_ve <= 1e-3 * max(1.0, rv.abs().max().item()), e_ve
    del m, w
    torch.cuda.empty_cache()


def test_full_depth_stream_prefix_measured_deltas():
    """(e) the FULL model (26 tower layers + projector + 28 decoder layers, vocab 152 064) in bf16 against the oracle on the same weights in fp32
    and in bf16: system prompt, 3 frame steps (one frame, then a 2-frame chunk), a query, 8 greedy tokens.  Records max |delta head logit| of this
    build and of the bf16 oracle, both against the fp32 oracle -- the measurement DESIGN.md section 2 quotes.
    tolerance: ours <= 2 x (bf16 oracle's own distance to fp32) + 3e-2; token ids equal the bf16 or the fp32 oracle's, or the fp32 top-2 margin at the
    first differing step is below 4 x the logit error (tie-fragile greedy on random weights, SURVEY.md section 7 hard part 5)."""
    m, w, ocfg = _build(28, 26, torch.bfloat16, vocab=152064, max_vit_batch=4, max_step_tokens=512)
    o32, o16 = _oracle(w, ocfg, torch.float32), _oracle(w, ocfg, torch.bfloat16)
    dev = m.device
    g = torch.Generator(device=dev).manual_seed(11)
    px = torch.randn(3, 3, 384, 384, generator=g, device=dev).to(torch.bfloat16)
    ve = m.visual_embed(px)
    v32, v16 = o32.visual_embed(px.float()), o16.visual_embed(px)
    res = dict(visual_embed=dict(ours_vs_fp32=maxerr(ve, v32), bf16_oracle_vs_fp32=maxerr(v16, v32), scale=v32.abs().max().item()))
    ids = torch.randint(0, 152064, (1, 40), generator=g, device=dev)
    emb = m.get_input_embeddings()
    steps = [('prompt+frame0', 'cat0'), ('frames1-2', 'cat12'), ('query', 'ids'), ]
    qids = torch.randint(0, 152064, (1, 24), generator=g, device=dev)

    def run(model, frames, dt):
        e = lambda i: model.get_input_embeddings()(i).to(dt)
        out, cache, logs = None, None, []
        for x in (torch.cat([e(ids), frames[:49][None].to(dt)], 1), frames[49:][None].to(dt), e(qids)):
            out = model(inputs_embeds=x, past_key_values=cache); cache = out.past_key_values
            rows = [x.shape[1] - 1] if x.shape[1] != 98 else [48, 97]
            logs.append(torch.cat([out.informative_logits[0, rows], out.relevance_logits[0, rows]], -1).float().cpu())
        return out, cache, logs

    out_h, cache_h, logs_h = run(m, ve, torch.bfloat16)
    out_32, cache_32, logs_32 = run(o32, ve.float(), torch.float32)            # the fp32 oracle is fed THIS build's frame embeddings: isolates the LLM side
    out_16, cache_16, logs_16 = run(o16, ve, torch.bfloat16)
    d_ours = max(maxerr(a, b) for a, b in zip(logs_h, logs_32))
    d_ref = max(maxerr(a, b) for a, b in zip(logs_16, logs_32))
    res['head_logits'] = dict(ours_vs_fp32=d_ours, bf16_oracle_vs_fp32=d_ref, ours_vs_bf16_oracle=max(maxerr(a, b) for a, b in zip(logs_h, logs_16)),
                              logit_scale=max(l.abs().max().item() for l in logs_32))
    # end to end incl. the tower: oracle fp32 on ITS OWN frame embeddings
    _, _, logs_e2e = run(o32, v32, torch.float32)
    res['head_logits_end_to_end'] = dict(ours_vs_fp32=max(maxerr(a, b) for a, b in zip(logs_h, logs_e2e)))
    # 8 greedy tokens from the generation prompt
    gen = torch.randint(0, 152064, (1, 5), generator=g, device=dev)
    from mmduet_amd.modeling_live import fast_greedy_generate
    buf = torch.zeros(1, 8, dtype=torch.long, device=dev)
    ids_h, _, _ = fast_greedy_generate(model=m, inputs_embeds=emb(gen), past_key_values=cache_h, eos_token_id=-1, inplace_output_ids=buf)
    ids_h = ids_h[0].tolist()
    toks = {}
    for name, om, cache, dt in (('fp32', o32, cache_32, torch.float32), ('bf16', o16, cache_16, torch.bfloat16)):
        b2 = torch.zeros(1, 8, dtype=torch.long, device=dev)
        t, _, _ = O.fast_greedy_generate(model=om, inputs_embeds=om.get_input_embeddings()(gen).to(dt), past_key_values=cache, eos_token_id=-1, inplace_output_ids=b2)
        toks[name] = t[0].tolist()
    res['tokens'] = dict(ours=ids_h, fp32_oracle=toks['fp32'], bf16_oracle=toks['bf16'])
    # lm_head logits of the first generated position and the fp32 top-2 margin
    lo = m(inputs_embeds=emb(gen), past_key_values=cache_h).logits[0, -1].float()
    l32 = o32(inputs_embeds=o32.get_input_embeddings()(gen), past_key_values=cache_32).logits[0, -1].float()
    top2 = l32.topk(2).values
    res['lm_logits'] = dict(ours_vs_fp32=maxerr(lo, l32), fp32_top2_margin=(top2[0] - top2[1]).item(), scale=l32.abs().max().item())
    _record('full_depth_bf16', **res)
    assert d_ours <= 2 * d_ref + 3e-2, res['head_logits']
    same = ids_h == toks['bf16'] or ids_h == toks['fp32']
    if not same:
        first = next(i for i in range(8) if ids_h[i] != toks['fp32'][i])
        assert first > 0 or res['lm_logits']['fp32_top2_margin'] < 4 * res['lm_logits']['ours_vs_fp32'], res
    del m, w, o32, o16
    torch.cuda.empty_cache()


# ---- the reference's autocast (fp16) tower --------------------------------------------------------------------------------------------------
def _rms(a, b):
    return (a.float() - b.float().to(a.device)).pow(2).mean().sqrt().item()


def test_fp16_tower_true_width_error_against_fp32():
    """config.tower_dtype = 'fp16' (models/modeling_live.py:28 runs the tower under torch.cuda.amp.autocast()): IEEE-half tower weights /
    activations (v_mfma_f32_16x16x32_f16), fp32 LayerNorm / softmax statistics, bf16 features out.  True width, 4 tower layers, 8 frames, fused-preprocess and
    pixel_values entry points.  Measured: rms error of the tower output against the fp32 oracle for the bf16 tower, the fp16 tower and the oracle's restatement of the
    autocast path; the fp16 tower must cut the bf16 tower's error at least in half and stay within 3 x the autocast oracle's own error (+ 1e-3 x scale)."""
    mb, w, ocfg = _build(1, 4, torch.bfloat16, max_vit_batch=8, tower_dtype='bf16')
    mh, _, _ = _build(1, 4, torch.bfloat16, max_vit_batch=8)          # the default: half wherever the kernels exist
    assert mb.tower_dtype == 'bf16' and mh.tower_dtype == 'fp16'
    dev = mb.device
    g = torch.Generator(device=dev).manual_seed(21)
    px = torch.randn(8, 3, 384, 384, generator=g, device=dev).to(torch.bfloat16)
    w32 = {k: v.float() for k, v in w.items()}
    t32 = O.vit_forward(w32, ocfg, px.float())
    tac = O.vit_forward_autocast_fp16(w, ocfg, px)
    tb16 = O.vit_forward(w, ocfg, px)
    fb, fh = mb.tower_features(px), mh.tower_features(px)
    assert fh.dtype == torch.bfloat16 and torch.isfinite(fh.float()).all()
    scale = t32.abs().max().item()
    res = dict(scale=scale, bf16_tower_rms=_rms(fb, t32), fp16_tower_rms=_rms(fh, t32), autocast_oracle_rms=_rms(tac, t32), bf16_oracle_rms=_rms(tb16, t32),
               bf16_tower_max=maxerr(fb, t32), fp16_tower_max=maxerr(fh, t32), autocast_oracle_max=maxerr(tac, t32))
    # the hidden state between the fp16 matmuls is fp32, as autocast's promotion makes it (VERDICT r03 "missing" 3): against the autocast restatement ITSELF the
    # default tower must sit closer than the round-3 form that rounds the stream to fp16 after every sublayer, and as close to fp32 as the restatement is
    mr, _, _ = _build(1, 4, torch.bfloat16, max_vit_batch=8, tower_dtype='fp16_resid16')
    assert mr.tower_dtype == 'fp16_resid16'
    fr16 = mr.tower_features(px)
    res.update(fp16_resid16_tower_rms=_rms(fr16, t32), fp16_tower_vs_autocast_oracle_rms=_rms(fh, tac), fp16_resid16_tower_vs_autocast_oracle_rms=_rms(fr16, tac))
    del mr
    assert res['fp16_tower_vs_autocast_oracle_rms'] <= res['fp16_resid16_tower_vs_autocast_oracle_rms'], res
    assert res['fp16_tower_rms'] <= 1.25 * res['autocast_oracle_rms'] + 1e-4 * scale, res
    # after projector + pooling (what the LLM sees)
    e32 = O.visual_embed(w32, ocfg, px.float())
    eb, eh = mb.visual_embed(px), mh.visual_embed(px)
    res.update(embed_scale=e32.abs().max().item(), bf16_tower_embed_rms=_rms(eb, e32), fp16_tower_embed_rms=_rms(eh, e32),
               autocast_oracle_embed_rms=_rms(O.visual_embed(w, ocfg, px, tower_autocast_fp16=True), e32))
    # the fused uint8 -> patch-matrix entry point gives the same tower as preprocess + visual_embed, bit for bit
    fr = torch.randint(0, 256, (3, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).to(dev)
    pv = mh.get_vision_tower().image_processor.preprocess(fr)['pixel_values']
    assert torch.equal(mh.visual_embed_frames(fr), mh.visual_embed(pv))
    _record('fp16_tower_4_layers', **res)
    assert res['fp16_tower_rms'] <= 0.5 * res['bf16_tower_rms'], res
    assert res['fp16_tower_rms'] <= 3 * res['autocast_oracle_rms'] + 1e-3 * scale, res
    assert res['fp16_tower_embed_rms'] <= 0.6 * res['bf16_tower_embed_rms'], res
    del mb, mh, w, w32
    torch.cuda.empty_cache()


def test_fp16_tower_full_depth_head_logit_delta():
    """The same at FULL depth (26 tower + 28 decoder layers): head logits of prompt + 3 frames end to end against the fp32 oracle, bf16 tower vs fp16 tower (the
    decoder is identical).  Recorded for DESIGN.md section 2; the fp16 tower's end-to-end delta must not exceed the bf16 tower's by more than 1e-2."""
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(11)
    px = torch.randn(3, 3, 384, 384, generator=g, device=dev).to(torch.bfloat16)
    ids = torch.randint(0, 152064, (1, 40), generator=g, device=dev)
    out = {}
    ref = None
    for name, td in (('bf16_tower', 'bf16'), ('fp16_tower', 'fp16')):
        m, w, ocfg = _build(28, 26, torch.bfloat16, vocab=152064, max_vit_batch=4, max_step_tokens=512, tower_dtype=td)
        if ref is None:
            o32 = _oracle(w, ocfg, torch.float32)
            v32 = o32.visual_embed(px.float())
            x = torch.cat([o32.get_input_embeddings()(ids), v32[None]], 1)
            r = o32(inputs_embeds=x)
            rows = [40 + 49 * (j + 1) - 1 for j in range(3)]
            ref = (v32, torch.cat([r.informative_logits[0, rows], r.relevance_logits[0, rows]], -1).float().cpu(), rows)
            del o32, r, x
        ve = m.visual_embed(px)
        o = m(inputs_embeds=torch.cat([m.get_input_embeddings()(ids), ve[None]], 1))
        lg = torch.cat([o.informative_logits[0, ref[2]], o.relevance_logits[0, ref[2]]], -1).float().cpu()
        out[name] = dict(visual_embed_rms=_rms(ve, ref[0]), visual_embed_max=maxerr(ve, ref[0]), head_logit_max_delta=maxerr(lg, ref[1]), embed_scale=ref[0].abs().max().item())
        del m, w, o
        torch.cuda.empty_cache()
    _record('fp16_tower_full_depth', **out)
    assert out['fp16_tower']['visual_embed_rms'] <= out['bf16_tower']['visual_embed_rms'], out
    assert out['fp16_tower']['head_logit_max_delta'] <= out['bf16_tower']['head_logit_max_delta'] + 1e-2, out


# ---- multi-GPU: collectives --------------------------------------------------------------------------------------------------------
def test_native_score_gather_world1():
    """mmd_gather_scores (RCCL bound at run time by libmmduet_hip) on a one-rank communicator: the all-gather degenerates to a copy of the padded block."""
    from mmduet_amd.distributed import NativeScoreGather
    dev = torch.device('cuda', 0)
    ng = NativeScoreGather(dev, rank=0, world=1)
    s = torch.rand(37, 2, device=dev)
    allsc, lens = ng.gather(s, 50)
    torch.cuda.synchronize()
    assert lens.tolist() == [37] and torch.equal(allsc[0, :37], s) and torch.isnan(allsc[0, 37:]).all()
    a2, l2 = ng.gather(s[:0], 50)
    torch.cuda.synchronize()
    assert l2.tolist() == [0] and torch.isnan(a2).all()
    # several streams per rank (mmd_gather_block) == the torch.distributed transport's layout; issued on a SIDE stream: the gather follows torch's current stream
    from mmduet_amd.distributed import gather_scores
    streams = [torch.rand(t, 2) for t in (50, 13, 0)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        a3, l3 = ng.gather_streams(streams, 50, 4)
    side.synchronize()
    ar, lr = gather_scores(streams, t_max=50, n_max=4)
    assert torch.equal(l3.cpu(), lr.cpu()) and torch.equal(torch.nan_to_num(a3.cpu()), torch.nan_to_num(ar.cpu()))
    ng.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs (the pool boxes have one; the 8-GPU driver run covers it)')
def test_bench_launches_its_own_ranks_nccl():
    """`python bench.py --gpus 2` with no torchrun environment spawns two ranks over RCCL and prints n_gpus = rccl_ranks = 2."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--tiny', '--frames', '12', '--steps', '1', '--warmup', '1', '--multi-stream', '0',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['config']['native_gather_check'] == 'ok'
