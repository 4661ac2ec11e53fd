#!/usr/bin/env python
"""KV-cache growth sweep (BASELINE configs[2]): LLM chunk-step time vs context length, up to the HBM limit -- ONE arena from 0 to the end.

The arena is the virtual-address reservation of mmd_stream_create (4 Mi tokens per row by default; MMDUET_KV_VIRTUAL_TOKENS raises it): growth maps pages behind
the same addresses, so nothing is reallocated, copied or restarted at any size.  Two kinds of points:
  * "real": the context is extended by REAL forwards (26-frame chunks of random embeddings, tower skipped) -- from 0 to ~131 k tokens in one go (the arena starts
    with 32 k tokens mapped and grows under the forwards), and again across the 1 M / 2 M / 4 M token marks;
  * "jump": between those stretches the context is advanced with mmd_kv_debug_set_len (slots declared live without computing them: same pages mapped, same
    attention work and traffic for the next step, no hours of prefill).
Writes gpurun_out/kv_growth_sweep.json."""
import ctypes as C, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import bench
from mmduet_amd._lib import lib, check

os.environ.setdefault('MMDUET_KV_VIRTUAL_TOKENS', str(5 << 20))
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
k = 26
frames = (torch.randn(k * 49, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
rows = [49 * (j + 1) - 1 for j in range(k)]
cache = model.new_cache(initial_tokens=32768)
arena = cache.arena
out = []


def real_steps(n, kind):
    """n real chunk forwards from the arena's current length; records the first and the last."""
    global cache
    for i in range(n):
        n0 = arena.length()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc, cache = model.frame_step(frames[None], type(cache)(arena, n0), rows)
        dt = time.perf_counter() - t0
        assert torch.isfinite(sc).all() and arena.length() == n0 + k * 49
        if i in (0, n - 1):
            out.append(dict(kind=kind, n_ctx=n0, kv_GB=round((n0 + k * 49) * 57344 / 1e9, 2), ms_per_26_frames=round(dt * 1e3, 2), llm_frames_per_s=round(k / dt, 1),
                            tokens_mapped=int(lib().mmd_kv_capacity(arena.h)), row_stride_tokens=int(lib().mmd_kv_stride(arena.h))))
            print(out[-1], flush=True)


try:
    real_steps(104, 'real: 0 -> 132 k tokens by forwards, arena growing from 32 k mapped')
    for mark in (300_000, 1_000_000, 2_000_000, 3_000_000, 4_000_000, 4_500_000):
        start = mark - 4 * k * 49
        check(lib().mmd_kv_debug_set_len(arena.h, start), model._ctx, 'set_len')          # jump (same arena)
        real_steps(8, f'real: forwards across the {mark:,}-token mark (after a jump)')
except Exception as e:
    out.append(dict(error=str(e)[:200], at_tokens=int(arena.length())))
    print(out[-1], flush=True)
free, total = torch.cuda.mem_get_info(dev)
out.append(dict(hbm_used_GB=round((total - free) / 1e9, 1), hbm_total_GB=round(total / 1e9, 1), final_tokens=int(arena.length())))
print(out[-1], flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'kv_growth_sweep.json'), 'w'), indent=1)
