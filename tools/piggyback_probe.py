#!/usr/bin/env python
"""What does ONE forward carrying a frame (49 rows, arena A) plus one decode row (arena B) cost at a given context, next to a plain frame step and a plain decode step?
    python tools/piggyback_probe.py [context=15000] [weights=fp8|bf16]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd._lib import lib, check
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
weights = sys.argv[2] if len(sys.argv) > 2 else 'fp8'
sys.argv = [sys.argv[0]]
args = bench.parse(['--weights', weights]); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
H = cfg.hidden_size
A = model.new_cache(initial_tokens=nctx + 8192); B = model.new_cache(initial_tokens=nctx + 8192)
for c in (A, B):
    check(lib().mmd_kv_debug_set_len(c.arena.h, nctx), model._ctx, 'set_len')
mk = lambda c: type(c)(c.arena, nctx)
frame = (torch.randn(49, H, device=dev) * 0.5).to(torch.bfloat16)
row = (torch.randn(1, H, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


t_frame = timeit(lambda: model.frame_step(frame[None], mk(A), [48]))
t_dec = timeit(lambda: model(inputs_embeds=row[None], past_key_values=mk(B)).logits)
for k in (1, 2, 4):
    x = frame.repeat(k, 1)
    t_fr_k = timeit(lambda: model.frame_step(x[None], mk(A), [49 * (j + 1) - 1 for j in range(k)]))
    t_both = timeit(lambda: model.multi_step([dict(x=x, cache=mk(A), head_rows=[49 * (j + 1) - 1 for j in range(k)]), dict(x=row, cache=mk(B), hidden='last')]))
    print(f'context {nctx} {weights}: {k} frame(s) alone {t_fr_k:.3f} ms | decode row alone {t_dec:.3f} ms | {k} frame(s) + decode row in ONE forward {t_both:.3f} ms  (sum of the two {t_fr_k + t_dec:.3f})', flush=True)
