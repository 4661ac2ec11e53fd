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


def last_form():
    f = (C.c_int * 2)(); lib().mmd_op_attention_last_form(model._ctx, f)
    return dict(form=FORMS.get(f[0], str(f[0])), splits_or_shared_blocks=f[1])


FORMS = {3: 'attn_gqa128_kernel<1> (decode ring)', 4: 'attn_gqa128_kernel<2> (grid form)', 5: 'attn_gqa128_w1_kernel', 8: 'attn_gqa128_chunk_kernel', 9: 'decode rows of several streams'}
one_frame = frames[:49]
one_row = frames[:1]


def small_steps(kind):
    """At the arena's current length: a 49-row frame step (the reference's own schedule, test/inference.py:239) and a decode row, timed (median of 5) and rolled back."""
    n0 = arena.length()
    rec = dict(kind=kind, n_ctx=n0)
    for name, x, rows in (('frame_step_49_rows', one_frame, [48]), ('decode_row', one_row, [0])):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sc, _c = model.frame_step(x[None], type(cache)(arena, n0), rows)
            ts.append(time.perf_counter() - t0)
            assert torch.isfinite(sc).all()
        rec[name + '_ms'] = round(sorted(ts[1:])[2] * 1e3, 3)
        rec[name + '_attention'] = last_form()
    arena.truncate(n0)
    out.append(rec); print(rec, flush=True)


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
                            tokens_mapped=int(lib().mmd_kv_capacity(arena.h)), row_stride_tokens=int(lib().mmd_kv_stride(arena.h)), attention=last_form()))
            print(out[-1], flush=True)


try:
    real_steps(104, 'real: 0 -> 132 k tokens by forwards, arena growing from 32 k mapped')
    small_steps('per-frame step and decode row at 132 k tokens')
    for mark in (300_000, 1_000_000, 2_000_000, 2_940_000, 4_000_000, 4_500_000):          # (2.94 M tokens = the 60 k-frame point of SURVEY section 8d config 3)
        start = mark - 4 * k * 49
        check(lib().mmd_kv_debug_set_len(arena.h, start), model._ctx, 'set_len')          # jump (same arena)
        real_steps(8, f'real: forwards across the {mark:,}-token mark (after a jump)')
        small_steps(f'per-frame step and decode row behind the {mark:,}-token mark')
except Exception as e:
    out.append(dict(error=str(e)[:200], at_tokens=int(arena.length())))
    print(out[-1], flush=True)
free, total = torch.cuda.mem_get_info(dev)
out.append(dict(hbm_used_GB=round((total - free) / 1e9, 1), hbm_total_GB=round(total / 1e9, 1), final_tokens=int(arena.length())))
print(out[-1], flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'kv_growth_sweep.json'), 'w'), indent=1)
