#!/usr/bin/env python
"""frames_per_forward sweep on a benchmark stream (one model build): frames/s, forwards, replayed frames; overlap on/off.
    python tools/k_sweep.py 13,26,52,78,104 [config] [overlap-only]        config: stream300 (default) | ground600 | qvh | youcook2 -> gpurun_out/k_sweep_<config>.json"""
import sys, os, time, json, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1,4,8,13,20,26,32,40').split(',')]
config = sys.argv[2] if len(sys.argv) > 2 else 'stream300'
overlaps = (True,) if len(sys.argv) > 3 else (True, False)
sys.argv = [sys.argv[0]]
args = bench.parse(['--config', config]); args.multi_stream = 0; args.frames_per_forward = max(ks)
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
query = 'Please narrate the video in real time.'[:24]
forced = sorted(random.Random(0).sample(range(1, args.frames + 1), args.responses)) if args.responses > 0 else []
out = []
for overlap in overlaps:
    for k in ks:
        args.frames_per_forward = k; args.no_overlap = not overlap
        d = bench.make_driver(args, model, tok, 1.0, forced)
        bench.run_stream(d, frames, query)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 2 if k > 2 else 1
        for _ in range(n): bench.run_stream(d, frames, query)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        cls = None
        if os.environ.get('K_SWEEP_CLASSES'):
            model.prof_reset(); model.prof_enable(True); bench.run_stream(d, frames, query); torch.cuda.synchronize(); model.prof_enable(False)
            cls = {kk: round(v['ms'], 1) for kk, v in model.prof_read().items() if v['ms'] > 0.5}
        rec = dict(classes=cls, frames_per_forward=k, overlap=overlap, frames_per_s=round(args.frames / dt, 1), ms_per_stream=round(dt * 1e3, 1), llm_forwards=d.forward_calls, replayed_frames=d.replayed_frames)
        out.append(rec); print(json.dumps(rec), flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', f'k_sweep_{config}.json'), 'w'), indent=1)
