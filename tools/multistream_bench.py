#!/usr/bin/env python
"""Throughput of S streams per GPU in shared forwards (mmduet_amd/multistream.py) against the single-stream schedule."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
cfgs = [(int(a), int(b)) for a, b in (x.split('x') for x in (sys.argv[1] if len(sys.argv) > 1 else '2x13,4x8,4x13,8x6').split(','))]
sys.argv = [sys.argv[0]]
args = bench.parse()
args.multi_stream = max(s for s, _ in cfgs); args.multi_frames_per_forward = max(k for _, k in cfgs)
args.multi_stream, args.multi_frames_per_forward = max(((s, k) for s, k in cfgs), key=lambda t: t[0] * (t[1] * 49 + 192))
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
query = 'Please narrate the video in real time.'[:24]
import random
forced = sorted(random.Random(0).sample(range(1, args.frames + 1), args.responses))
d = bench.make_driver(args, model, tok, 1.0, forced)
bench.run_stream(d, frames, query)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2): bench.run_stream(d, frames, query)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print(f'single stream k={args.frames_per_forward}: {args.frames / dt:.1f} frames/s ({dt * 1e3:.0f} ms/stream)', flush=True)
for S, k in cfgs:
    fps, ms, rounds, replay, frac = bench.run_multi_stream(args, model, tok, frames, query, S, k, steps=2, warmup=1, device=dev)
    print(f'{S} streams x k={k}: {fps:.1f} frames/s ({ms:.0f} ms per {S}-stream step, {ms / S:.0f} ms/stream, {rounds} rounds, {replay} replayed frames, {100 * frac:.0f}% of wall inside the merged forwards)', flush=True)
# per-class GPU time (HIP events around every launch; serialises nothing but adds event overhead): where does a step go?
def classes(fn):
    model.prof_reset(); model.prof_enable(True); fn(); torch.cuda.synchronize(); model.prof_enable(False)
    p = model.prof_read()
    return {k: round(v['ms'], 1) for k, v in p.items() if v['ms'] > 0.5}
print('single-stream classes (ms):', classes(lambda: bench.run_stream(d, frames, query)), flush=True)
S, k = cfgs[-1]
print(f'{S}x{k} classes (ms):', classes(lambda: bench.run_multi_stream(args, model, tok, frames, query, S, k, steps=1, warmup=0, device=dev)), flush=True)
