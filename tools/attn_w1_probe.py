#!/usr/bin/env python
"""attn_gqa128_w1_kernel (variant 5) / attn_gqa128_chunk_kernel (variant 6: python tools/attn_w1_probe.py chunk) against fp32 math and against the shipped forms (variant 3): correctness on the production shapes, then wall per shape
(mmd_op_attention_bench: constant data, attention + merge).  python tools/attn_w1_probe.py [quick]"""
import ctypes as C, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from mmduet_amd._lib import lib, check
from rawops import RawOps
ops = RawOps(torch.bfloat16)
dev = ops.dev
nh, nkv, d = 28, 4, 128

def ref_attention(q, K, V, n_ctx):
    S = q.shape[0]; G = nh // nkv
    qf = q.float().view(S, nh, d); out = torch.empty(S, nh, d, device=dev)
    pos = n_ctx + torch.arange(S, device=dev)
    for h in range(nh):
        k = K[h // G, :n_ctx + S].float(); v = V[h // G, :n_ctx + S].float()
        s = qf[:, h] @ k.T / d ** 0.5
        s = s.masked_fill(torch.arange(n_ctx + S, device=dev)[None, :] > pos[:, None], float('-inf'))
        out[:, h] = torch.softmax(s, -1) @ v
    return out.view(S, nh * d)

bad = 0
VNEW = 6 if (len(sys.argv) > 1 and sys.argv[1] == 'chunk') else 5
shapes = [(49, 15000), (49, 0), (49, 1), (49, 63), (49, 64), (49, 1024), (49, 30000), (24, 15000), (64, 4096), (98, 15000), (196, 8000), (131, 15000), (147, 0), (146, 0), (150, 63), (183, 64),
          (200, 100), (1274, 0), (1274, 15000), (1323, 8000), (637, 3), (2, 70001), (10, 5), (17, 300), (37, 129), (300, 70000), (512, 4097)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick': shapes = shapes[:8]
if VNEW == 6: shapes = [(1274, 0), (1274, 15000), (1274, 30000), (1323, 8000), (147, 0), (147, 1), (150, 63), (183, 64), (200, 100), (300, 70000), (637, 3), (1911, 27000), (2058, 127), (512, 4097), (37, 129), (392, 15000), (40, 5000)]
for S, n in shapes:
    g = torch.Generator(device=dev).manual_seed(S + n)
    cap = (n + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nkv, cap, d, generator=g, device=dev).to(torch.bfloat16)
    K[:, n + S:] = 1e4; V[:, n + S:] = 1e4
    ref = ref_attention(q, K, V, n)
    o5 = ops.attention(q, K, V, nh, nkv, d, n, True, VNEW).float()
    o3 = ops.attention(q, K, V, nh, nkv, d, n, True, 3).float()
    e5 = ((o5 - ref).abs().max() / max(1.0, ref.abs().max().item())).item(); e3 = ((o3 - ref).abs().max() / max(1.0, ref.abs().max().item())).item()
    rep = all(torch.equal(ops.attention(q, K, V, nh, nkv, d, n, True, VNEW).float(), o5) for _ in range(5))
    ok = bool(torch.isfinite(o5).all()) and e5 <= 1.8e-2 and rep
    bad += not ok
    print(f'S={S:5d} n={n:6d}  w1 err {e5:.2e}  shipped err {e3:.2e}  repeat-identical {rep}  {"ok" if ok else "FAIL"}', flush=True)
print('correctness:', 'ALL OK' if not bad else f'{bad} FAILED', flush=True)
for S in ((24, 49, 64, 98, 196, 392, 1274) if VNEW == 5 else (147, 392, 637, 1274, 1911)):
    for n in (0, 1024, 4096, 15000, 30000):
        r = {}
        for v in (3, VNEW):
            ms = C.c_float()
            check(lib().mmd_op_attention_bench(ops.ctx, S, nh, nkv, d, n, v, 20, C.byref(ms)), ops.ctx)
            r[v] = ms.value * 1e3
        fl = 4.0 * S * (n + S) * 128 * 28
        print(f'S={S:4d} n={n:6d}  variant 3 {r[3]:8.1f} us  variant {VNEW} {r[VNEW]:8.1f} us  ({fl / r[VNEW] / 1e6:7.1f} TF/s)  x{r[3] / r[VNEW]:.2f}', flush=True)
