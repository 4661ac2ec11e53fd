#!/usr/bin/env python
"""Per-class kernel time of ONE LLM step of S rows at a given context (every launch bracketed).   python tools/step_classes_probe.py [context] [weights] [S,S,...]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd._lib import lib, check
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
weights = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
Ss = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '1,5,24,49,98,196,637,1274').split(',')]
sys.argv = [sys.argv[0]]
args = bench.parse(['--weights', weights]); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
A = model.new_cache(initial_tokens=nctx + 8192)
check(lib().mmd_kv_debug_set_len(A.arena.h, nctx), model._ctx, 'set_len')
for S in Ss:
    x = (torch.randn(1, S, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
    f = lambda: model(inputs_embeds=x, past_key_values=type(A)(A.arena, nctx))
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
    model.prof_reset(); model.prof_set_stride(1); model.prof_enable(True); f(); torch.cuda.synchronize(); model.prof_enable(False)
    p = model.prof_read()
    print(f'S={S:5d} ctx={nctx} {weights}: wall {wall:7.3f} ms | ' + ' '.join(f'{k} {v["ms"]:.2f}ms/{v["launches"]}' for k, v in p.items() if v['launches']), flush=True)
