#!/usr/bin/env python
"""ViT attention (attn_rowmajor) and chunk attention (attn_gqa128) timed INSIDE the model on random data: per-launch HIP-event time of the class, TF/s, fraction of 2.5 PF.
    python tools/vit_attn_bench.py [rounds]   (env switches of the kernels are read per process: run once per variant)"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sys.argv = [sys.argv[0]]
args = bench.parse(); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
px = torch.randn(35, 3, 384, 384, device=dev).to(torch.bfloat16)
model.visual_embed(px); torch.cuda.synchronize()
res = {}
ts = []
for r in range(rounds):
    model.prof_reset(); model.prof_set_stride(1); model.prof_enable(['attn_vit'])
    model.visual_embed(px); torch.cuda.synchronize(); model.prof_enable(False)
    p = model.prof_read()['attn_vit']; ts.append(p['ms'] / p['launches'] * 1e3)
fl = 4.0 * 35 * 729 * 729 * 1152
us = sorted(ts)[len(ts) // 2]
res['vit_attention_35_frames'] = dict(us_per_layer=round(us, 1), tflops=round(fl / us / 1e6, 1), frac_of_2500=round(fl / us / 1e6 / 2500, 3))
print('ViT attention, 35 frames x 16 heads x 729^2 x 72:', res['vit_attention_35_frames'], flush=True)
if os.environ.get('VIT_ONLY'): sys.exit(0)
# chunk attention: 26-frame chunks at growing context
x = (torch.randn(1, 1274, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
cache = None
for n_chunks in range(12):
    model.prof_reset(); model.prof_set_stride(1); model.prof_enable(['attn_llm'])
    out = model(inputs_embeds=x, past_key_values=cache); cache = out.past_key_values
    torch.cuda.synchronize(); model.prof_enable(False)
    p = model.prof_read()['attn_llm']
    n_ctx = n_chunks * 1274
    fl = 4.0 * 28 * 128 * 1274 * (n_ctx + 1274 / 2.0)          # causal: half of the chunk's own block
    us = p['ms'] / p['launches'] * 1e3
    if n_chunks in (0, 3, 7, 11):
        res[f'chunk_attention_ctx{n_ctx}'] = dict(us_per_layer=round(us, 1), tflops=round(fl / us / 1e6, 1), frac_of_2500=round(fl / us / 1e6 / 2500, 3), launches=int(p['launches']))
        print(f'chunk attention S=1274 over {n_ctx} keys:', res[f'chunk_attention_ctx{n_ctx}'], flush=True)
# reference point only (never on the product path): the ROCm library attention (torch SDPA -> its flash kernels) on the same problem sizes, random data.
# ViT: [35, 16, 729, 72] non-causal, fp16 and bf16.  Chunk: 28 query heads x 1274 rows over n + 1274 keys of 4 kv heads (enable_gqa), NO mask (the library's
# is_causal is top-left aligned; the unmasked problem is the same work as ours plus half of the chunk's own 1274 x 1274 block).
import torch.nn.functional as F
def timed(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
if os.environ.get('ATTN_LIBRARY', '1') != '0':
    for dt in (torch.float16, torch.bfloat16):
        q, k, v = (torch.randn(35, 16, 729, 72, device=dev).to(dt) for _ in range(3))
        try:
            us = timed(lambda: F.scaled_dot_product_attention(q, k, v))
            fl = 4.0 * 35 * 729 * 729 * 1152
            res[f'library_vit_attention_{str(dt)[6:]}'] = dict(us_per_layer=round(us, 1), tflops=round(fl / us / 1e6, 1), frac_of_2500=round(fl / us / 1e6 / 2500, 3))
        except Exception as ex:
            res[f'library_vit_attention_{str(dt)[6:]}'] = dict(error=str(ex)[:200])
        print('library SDPA, ViT shape', dt, res[f'library_vit_attention_{str(dt)[6:]}'], flush=True)
    for n_ctx in (0, 3822, 8918, 14014):
        q = torch.randn(1, 28, 1274, 128, device=dev).to(torch.bfloat16)
        k, v = (torch.randn(1, 4, n_ctx + 1274, 128, device=dev).to(torch.bfloat16) for _ in range(2))
        key = f'library_chunk_attention_ctx{n_ctx}_unmasked'
        try:
            us = timed(lambda: F.scaled_dot_product_attention(q, k, v, enable_gqa=True))
            fl = 4.0 * 28 * 128 * 1274 * (n_ctx + 1274)
            res[key] = dict(us_per_layer=round(us, 1), tflops=round(fl / us / 1e6, 1), frac_of_2500=round(fl / us / 1e6 / 2500, 3))
        except Exception as ex:
            res[key] = dict(error=str(ex)[:200])
        print('library SDPA, chunk shape over', n_ctx, 'keys:', res[key], flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(R, 'gpurun_out', f'attn_bench_{os.environ.get("ATTN_TAG", "default")}.json'), 'w'), indent=1)
