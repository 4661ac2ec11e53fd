#!/usr/bin/env python
"""Does confining the vision tower to a CU subset (hipExtStreamCreateWithCUMask) let token-by-token decoding run beside it?"""
import sys, os, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd.modeling_live import fast_greedy_generate
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)
px = torch.randn(32, 3, 384, 384, device=dev).to(torch.bfloat16)
ctx = (torch.randn(1, 980, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
prompt = (torch.randn(1, 13, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
def gen(ntok, cache):
    out = torch.zeros(1, ntok, dtype=torch.long, device=dev)
    return fast_greedy_generate(model=model, inputs_embeds=prompt, past_key_values=cache, eos_token_id=-1, inplace_output_ids=out)
def vit(n):
    for _ in range(n): model.visual_embed(px)
cache = None
for _ in range(6):
    cache = model(inputs_embeds=ctx, past_key_values=cache).past_key_values
base = cache
gen(8, model.cache_prefix(base, len(base))); vit(1); torch.cuda.synchronize()
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
tg = t(lambda: gen(32, model.cache_prefix(base, len(base)))); tv = t(lambda: vit(3))
print(f'alone: gen32 {tg:.1f} ms, vit3 {tv:.1f} ms', flush=True)
ALL = (1 << 256) - 1
main = torch.cuda.Stream(device=dev)            # decode on a NON-null stream: a CU-masked stream is 'blocking' and would serialise against the null stream
plain = torch.cuda.Stream(device=dev)
def both_plain(n):
    with torch.cuda.stream(plain): vit(n)
    with torch.cuda.stream(main): gen(32, model.cache_prefix(base, len(base)))
print(f'plain non-blocking side stream: gen32 || vit3 {t(lambda: both_plain(3)):.1f} ms, gen32 || vit6 {t(lambda: both_plain(6)):.1f} ms', flush=True)
masks = {'all256': ALL, 'low128': (1 << 128) - 1, 'low192': (1 << 192) - 1, 'low224': (1 << 224) - 1, 'high128': ((1 << 128) - 1) << 128,
         'even128': int('01' * 128, 2), 'blk8_128': int(('0' * 8 + '1' * 8) * 16, 2), 'blk16_128': int(('0' * 16 + '1' * 16) * 8, 2), 'blk32_128': int(('0' * 32 + '1' * 32) * 4, 2)}
for name, bits in masks.items():
    try:
        s = masked_stream(bits)
    except AssertionError as e:
        print(name, 'create failed', e); continue
    def vit_on():
        with torch.cuda.stream(s): vit(3)
    tvm = t(vit_on)
    def both():
        with torch.cuda.stream(s): vit(3)
        with torch.cuda.stream(main): gen(32, model.cache_prefix(base, len(base)))
    tb = t(both)
    def both2():
        with torch.cuda.stream(s): vit(6)
        with torch.cuda.stream(main): gen(32, model.cache_prefix(base, len(base)))
    tb2 = t(both2)
    print(f'{name:10s}: vit3 masked alone {tvm:.1f} ms | gen32 || vit3 {tb:.1f} ms (serial sum {tg + tv:.1f}) | gen32 || vit6 {tb2:.1f} ms (serial {tg + 2 * tv:.1f})', flush=True)
