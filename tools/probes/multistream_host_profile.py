#!/usr/bin/env python
"""Where the host time of a 4-stream shared-forward step goes (cProfile of every slot thread + the scheduler): tools/probes/multistream_host_profile.py [S] [k]"""
import sys, os, time, cProfile, pstats, io, threading
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch, bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
k = int(sys.argv[2]) if len(sys.argv) > 2 else 13
sys.argv = [sys.argv[0]]
args = bench.parse(); args.multi_stream, args.multi_frames_per_forward = S, k
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
query = 'Please narrate the video in real time.'[:24]
bench.run_multi_stream(args, model, tok, frames, query, S, k, steps=1, warmup=1, device=dev)
prof = cProfile.Profile()
threading.setprofile(lambda *a: None)
orig_run = threading.Thread.run
def run_profiled(self):
    p = cProfile.Profile(); p.enable()
    try: orig_run(self)
    finally:
        p.disable(); profs.append(p)
profs = []
threading.Thread.run = run_profiled
prof.enable()
t0 = time.perf_counter()
out = bench.run_multi_stream(args, model, tok, frames, query, S, k, steps=1, warmup=0, device=dev)
dt = time.perf_counter() - t0
prof.disable()
print('step', round(dt, 3), 's', out)
st = pstats.Stats(prof)
for p in profs: st.add(p)
s = io.StringIO(); st.stream = s
st.sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
