#!/usr/bin/env python
"""Decode-step probe: N greedy tokens at a given context length on the 7B model (context declared live with mmd_kv_debug_set_len: same traffic, no prefill).
    python tools/decode_probe.py [tokens=64] [context=15000] [weights=bf16|fp8]      prints ms per token; run under rocprofv3 --kernel-trace for the per-kernel picture"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd._lib import lib, check
from mmduet_amd.modeling_live import fast_greedy_generate
ntok = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
weights = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
sys.argv = [sys.argv[0]]
args = bench.parse(['--weights', weights]); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
cache = model.new_cache(initial_tokens=nctx + 4096)
check(lib().mmd_kv_debug_set_len(cache.arena.h, nctx), model._ctx, 'set_len')
prompt = (torch.randn(1, 5, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
for it in range(3):
    out = torch.zeros(1, ntok, dtype=torch.long)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fast_greedy_generate(model=model, inputs_embeds=prompt, past_key_values=type(cache)(cache.arena, nctx), eos_token_id=-1, inplace_output_ids=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'iter {it}: {ntok} tokens at context {nctx} ({weights}): {dt*1e3:.1f} ms = {dt*1e3/ntok:.3f} ms/token', flush=True)
