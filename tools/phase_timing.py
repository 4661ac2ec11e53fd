#!/usr/bin/env python
"""Wall-clock phase breakdown of one bench stream (sync after each phase) -- diagnostic only."""
import sys, os, time, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, args.resolution, args.resolution), dtype=torch.uint8).to(dev)
T = args.frames
forced = sorted(random.Random(0).sample(range(1, T + 1), args.responses))
d = bench.make_driver(args, model, tok, 1.0, forced)
gen_t = [0.0]
orig = d._generate_response
def timed_gen():
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig(); torch.cuda.synchronize(); gen_t[0] += time.perf_counter() - t; return r
d._generate_response = timed_gen
for it in range(2):
    gen_t[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.reset(); torch.cuda.synchronize(); t1 = time.perf_counter()
    d.input_video_stream(frames); torch.cuda.synchronize(); t2 = time.perf_counter()
    d.input_query_stream([{'role': 'user', 'content': 'narrate', 'time': 0.0}])
    d.inference(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'iter {it}: reset {1e3*(t1-t0):.1f} ms | preprocess+ViT {1e3*(t2-t1):.1f} ms | phase B total {1e3*(t3-t2):.1f} ms (generation {1e3*gen_t[0]:.1f} ms, frames {1e3*(t3-t2-gen_t[0]):.1f} ms) | forwards {d.forward_calls} replayed {d.replayed_frames}')
