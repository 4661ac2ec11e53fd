#!/usr/bin/env python
"""KV-cache growth sweep (BASELINE config 3): LLM frame-step time vs context length, up to the HBM limit.

Feeds random frame embeddings (vision tower skipped) in 20-frame causal chunks; between measurements the context is
advanced with mmd_kv_debug_set_len (slots declared live without computing them: same attention work and traffic)."""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from mmduet_amd._lib import lib, check

sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
k = 20
frames = (torch.randn(k * 49, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
cache = model.new_cache(initial_tokens=32768)
rows = [49 * (j + 1) - 1 for j in range(k)]
out = []
for n_ctx in (0, 29400, 100_000, 300_000, 1_000_000, 2_000_000, 3_000_000, 4_000_000, 4_500_000):
    try:
        if n_ctx >= 3_000_000:      # growth by reallocation needs old + new arena at once: start the big ones fresh
            import gc
            h = h2 = sc = None; del cache; gc.collect(); torch.cuda.synchronize()
            cache = model.new_cache(initial_tokens=n_ctx + 2048)
        check(lib().mmd_kv_debug_set_len(cache.arena.h, n_ctx), model._ctx, 'set_len')
    except Exception as e:
        out.append(dict(n_ctx=n_ctx, error=str(e)[:120])); break
    h = type(cache)(cache.arena, n_ctx)
    torch.cuda.synchronize(); ts = []
    for it in range(3):
        t0 = time.perf_counter()
        sc, h2 = model.frame_step(frames[None], type(cache)(cache.arena, n_ctx), rows)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    kv_gb = (n_ctx + k * 49) * 57344 / 1e9
    out.append(dict(n_ctx=n_ctx, kv_GB=round(kv_gb, 2), ms_per_20_frames=round(dt * 1e3, 2), llm_frames_per_s=round(k / dt, 1),
                    arena_capacity_tokens=int(lib().mmd_kv_capacity(cache.arena.h))))
    print(out[-1], flush=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'kv_growth_sweep.json'), 'w'), indent=1)
