#!/usr/bin/env python
"""Anatomy of the multi-stream response leg (mmduet_amd/multistream.py): per class of scheduler round (which segments shared the forward) the count, the time inside the
merged forward and the host time in front of it.  usage: multistream_anatomy.py 4x13[,8x6...] [videos_per_slot]"""
import sys, os, time, json, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
cfgs = [(int(a), int(b)) for a, b in (x.split('x') for x in (sys.argv[1] if len(sys.argv) > 1 else '4x13').split(','))]
per_slot = int(sys.argv[2]) if len(sys.argv) > 2 else 1
phase_b = len(sys.argv) > 3 and sys.argv[3] == 'b'          # Phase B alone: the frame embeddings are computed once up front and handed over as features (no tower in the timed pass)
sys.argv = [sys.argv[0]]
args = bench.parse(['--weights', os.environ['MS_WEIGHTS']] if os.environ.get('MS_WEIGHTS') else [])          # MS_WEIGHTS=fp8: e4m3 decoder weights (BASELINE configs[4])
args.multi_stream, args.multi_frames_per_forward = max(((s, k) for s, k in cfgs), key=lambda t: t[0] * (t[1] * 49 + 192))
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
frames = torch.randint(0, 256, (args.frames, 3, 336, 336), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
query = 'Please narrate the video in real time.'[:24]
if phase_b:
    with torch.no_grad():
        feats = torch.cat([model.visual_embed_frames(frames[b0:b0 + 35]) for b0 in range(0, args.frames, 35)]).view(args.frames, cfg.frame_num_tokens, -1)
    frames = feats
out = {}
for S, k in cfgs:
    r = bench.MultiRunner(args, model, tok, frames, query, S * per_slot, k)
    r.ms.n_slots = S
    if os.environ.get('MS_DYNAMIC'):
        r.ms.dynamic_chunks = True; r.ms.max_chunk_frames = int(os.environ['MS_DYNAMIC'])
    if os.environ.get('MS_LOOKAHEAD'):
        r.ms.vit_lookahead_batches = int(os.environ['MS_LOOKAHEAD'])
    r.run(); torch.cuda.synchronize()
    r.ms.round_log = []; r.ms.rounds = 0; r.ms.exec_seconds = 0.0
    t0 = time.perf_counter(); r.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    log = r.ms.round_log
    cls = collections.OrderedDict()
    prev_end = t0
    for kinds, rows, ts, te in log:
        key = ''.join(sorted(kinds))
        c = cls.setdefault(key, dict(n=0, exec_ms=0.0, host_ms=0.0, rows=0))
        c['n'] += 1; c['exec_ms'] += (te - ts) * 1e3; c['host_ms'] += (ts - prev_end) * 1e3; c['rows'] += sum(rows)
        prev_end = te
    fhist = collections.Counter(r for kinds, rows, ts, te in log for kk, r in zip(kinds, rows) if kk == 'f')
    print('   frame-segment rows (rows: count):', dict(sorted(fhist.items())), flush=True)
    fps = S * per_slot * args.frames / dt
    print(f'== {S} slots x k={k}, {S * per_slot} videos: {fps:.1f} frames/s, wall {dt * 1e3:.0f} ms, {len(log)} rounds, in forwards {r.ms.exec_seconds / dt * 100:.1f} %, replayed {sum(x["replayed_frames"] for x in r.last)}', flush=True)
    for key, c in sorted(cls.items(), key=lambda t: -t[1]['exec_ms'] - t[1]['host_ms']):
        print(f'   {key:10s} n={c["n"]:4d} rows/round={c["rows"] / c["n"]:7.1f} exec {c["exec_ms"]:8.1f} ms ({c["exec_ms"] / c["n"]:6.2f}/round) host-before {c["host_ms"]:7.1f} ms ({c["host_ms"] / c["n"]:5.2f}/round)')
    out[f'{S}x{k}'] = dict(frames_per_s=round(fps, 1), wall_ms=round(dt * 1e3, 1), rounds=len(log), in_forwards=round(r.ms.exec_seconds / dt, 3), replayed=sum(x['replayed_frames'] for x in r.last), classes={k2: {a: round(b, 2) for a, b in v.items()} for k2, v in cls.items()})
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'multistream_anatomy.json'), 'w'), indent=1)
