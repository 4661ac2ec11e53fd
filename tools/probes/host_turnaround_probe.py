"""Where does the host spend the ~250 us between two chunk forwards?  Wraps model.frame_step: time inside the call (native forward + the sync that fetches the head logits),
time between a return and the next call (the driver's Python), for the chunk forwards of one stream300 pass."""
import sys, os, time, statistics
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch, bench, random
sys.argv = [sys.argv[0]]
args = bench.parse(); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
forced = sorted(random.Random(0).sample(range(1, args.frames + 1), args.responses)) if args.responses > 0 else []
d = bench.make_driver(args, model, tok, 1.0, forced)
bench.run_stream(d, frames, 'Please narrate the video.')
inside, between, last = [], [], [None]
orig = model.frame_step
def wrapped(*a, **k):
    t0 = time.perf_counter()
    if last[0] is not None: between.append(t0 - last[0])
    r = orig(*a, **k)
    t1 = time.perf_counter(); inside.append(t1 - t0); last[0] = t1
    return r
model.frame_step = wrapped
d = bench.make_driver(args, model, tok, 1.0, forced)
torch.cuda.synchronize(); t0 = time.perf_counter(); bench.run_stream(d, frames, 'Please narrate the video.'); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'stream {dt*1e3:.1f} ms; {len(inside)} chunk forwards: inside median {statistics.median(inside)*1e3:.2f} ms (sum {sum(inside)*1e3:.1f}); between-calls median {statistics.median(between)*1e6:.0f} us, '
      f'sum {sum(between)*1e3:.1f} ms, sorted us {[round(b*1e6) for b in sorted(between)]}')
# split the between-time of no-response gaps: pure Python pieces
import cProfile, pstats, io
pr = cProfile.Profile(); d = bench.make_driver(args, model, tok, 1.0, forced); pr.enable(); bench.run_stream(d, frames, 'Please narrate the video.'); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:2600])
