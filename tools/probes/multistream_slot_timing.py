#!/usr/bin/env python
"""Host time of a watching slot between two merged forwards (tools/probes/multistream_slot_timing.py [S] [k]): from the moment `post` returns a round's results to the moment the
slot posts its next request, split into the tower issue (`_issue_vit`), the step input (`_step_input`) and the rest (decisions, bookkeeping, thread hand-off)."""
import sys, os, time, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd import multistream, inference
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
k = int(sys.argv[2]) if len(sys.argv) > 2 else 13
sys.argv = [sys.argv[0]]
args = bench.parse(); args.multi_stream, args.multi_frames_per_forward = S, k
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
query = 'Please narrate the video in real time.'[:24]
bench.run_multi_stream(args, model, tok, frames, query, S, k, steps=1, warmup=0, device=dev)
T = collections.Counter(); N = collections.Counter()
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **kw):
        t0 = time.perf_counter()
        try: return f(self, *a, **kw)
        finally: T[name] += time.perf_counter() - t0; N[name] += 1
    setattr(cls, name, g)
D = bench.bench_driver_class()
for n in ('_issue_vit', '_step_input', '_decide', '_chunk_size', '_prefix_ids_for_next_frame'):
    wrap(inference.LiveInferForBenchmark, n)
post0 = multistream._Slot.post
last = {}
def post(self, request):
    now = time.perf_counter()
    if self.index in last:
        T['between_posts_' + request.kind] += now - last[self.index]; N['between_posts_' + request.kind] += 1
    r = post0(self, request)
    last[self.index] = time.perf_counter()
    return r
multistream._Slot.post = post
t0 = time.perf_counter()
out = bench.run_multi_stream(args, model, tok, frames, query, S, k, steps=1, warmup=0, device=dev)
print('pass', round(time.perf_counter() - t0, 3), 's', out)
for key in sorted(T, key=lambda x: -T[x]):
    print(f'{key:34s} n={N[key]:5d} total {T[key] * 1e3:8.1f} ms  avg {T[key] / N[key] * 1e6:8.1f} us')
