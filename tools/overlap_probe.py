#!/usr/bin/env python
"""How well do the vision tower (MFMA-bound) and token-by-token decoding (HBM-bound) share the GPU on two HIP streams?"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd.modeling_live import fast_greedy_generate
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
px = torch.randn(32, 3, 384, 384, device=dev).to(torch.bfloat16)
ctx = (torch.randn(1, 980, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
prompt = (torch.randn(1, 13, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
side = torch.cuda.Stream(device=dev)
def gen(ntok, cache):
    out = torch.zeros(1, ntok, dtype=torch.long, device=dev)
    return fast_greedy_generate(model=model, inputs_embeds=prompt, past_key_values=cache, eos_token_id=-1, inplace_output_ids=out)
def vit(n):
    for _ in range(n): model.visual_embed(px)
def chunk(cache, n):
    for _ in range(n):
        cache = model(inputs_embeds=ctx, past_key_values=cache).past_key_values
    return cache
base = chunk(None, 6)           # ~6k tokens of context
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); gen(32, model.cache_prefix(base, len(base))); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); vit(3); torch.cuda.synchronize(); tv = time.perf_counter() - t0
    t0 = time.perf_counter(); chunk(model.cache_prefix(base, len(base)), 3); torch.cuda.synchronize(); tc = time.perf_counter() - t0
    # overlapped: tower on the side stream, decode on the main stream
    t0 = time.perf_counter()
    with torch.cuda.stream(side): vit(3)
    gen(32, model.cache_prefix(base, len(base))); torch.cuda.synchronize(); tgv = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(side): vit(3)
    chunk(model.cache_prefix(base, len(base)), 3); torch.cuda.synchronize(); tcv = time.perf_counter() - t0
    print(f'iter {it}: gen32 {tg*1e3:.1f} ms | vit3 {tv*1e3:.1f} ms | chunk3 {tc*1e3:.1f} ms | gen32 || vit3 {tgv*1e3:.1f} ms (sum {1e3*(tg+tv):.1f}) | chunk3 || vit3 {tcv*1e3:.1f} ms (sum {1e3*(tc+tv):.1f})')
