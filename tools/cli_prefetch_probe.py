#!/usr/bin/env python
"""The CLI (`python -m mmduet_amd`) over N synthetic Motion-JPEG clips with and without clip prefetch, 7B / so400m at true shapes (random weights): wall per run, records equal,
and -- under `rocprofv3 --kernel-trace` (tools/cli_prefetch_probe.sh) -- the GPU idle gaps between videos.   python tools/cli_prefetch_probe.py [workers] [n_clips] [frames]"""
import json, os, sys, time, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import bench
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 8
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 120
sys.argv = [sys.argv[0]]
args = bench.parse(); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
import mmduet_amd.inference as inf
import mmduet_amd.__main__ as cli
from mmduet_amd.video_decode import write_mjpeg_avi
inf.build_model_and_tokenizer = lambda **kw: (model, tok)
d = os.environ.get('CLI_PROBE_DIR') or tempfile.mkdtemp(prefix='mmduet_cli_')
os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(0)
entries = []
t0 = time.perf_counter()
for i in range(n_clips):
    path = os.path.join(d, f'clip{i}.avi')
    if not os.path.exists(path):
        base = rng.integers(0, 256, (1, 336, 448, 3), dtype=np.uint8)          # 2-fps source, twice as many frames as the 1-fps schedule keeps; smooth content + noise: realistic JPEG sizes
        fr = (base.astype(np.int16) + rng.integers(-24, 24, (2 * frames, 336, 448, 3), dtype=np.int16)).clip(0, 255).astype(np.uint8)
        write_mjpeg_avi(path, fr, 2.0, quality=85)
    entries.append({'question_id': f'q{i}', 'video': f'clip{i}.avi', 'conversation': [{'role': 'user', 'content': 'Please narrate the video in real time.', 'time': 0.0}]})
json.dump(entries, open(os.path.join(d, 'test.json'), 'w'))
t_make = time.perf_counter() - t0
flags = ['--live_version', 'test', '--llm_pretrained', 'synthetic:0', '--input_dir', d, '--test_fname', os.path.join(d, 'test.json'), '--frame_fps', '1', '--frame_resolution', '336',
         '--max_num_frames', str(frames), '--stream_end_prob_threshold', '1.0', '--frames_per_forward', '26', '--bf16', 'true']
# host-side phase times of the CLI's main loop (diagnostic: wrappers around the product's functions, product code unchanged)
phase = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); phase.setdefault(name, []).append((time.perf_counter() - t) * 1e3); return r
    return w
import mmduet_amd.prefetch as pfm, mmduet_amd.results as resm, mmduet_amd.video_input as vim
D = inf.LiveInferForBenchmark
for nm in ('reset', 'input_video_stream', 'input_query_stream', 'inference'):
    setattr(D, nm, timed(nm, getattr(D, nm)))
pfm.ClipPrefetcher.take = timed('prefetcher.take', pfm.ClipPrefetcher.take)
vim.load_video_frames = timed('load_video_frames (upload wait + letterbox)', vim.load_video_frames)
resm.result_record = timed('result_record', resm.result_record)
out = {}
for w in ([0, workers] if os.environ.get('CLI_PROBE_AB', '1') == '1' else [workers]):
    torch.cuda.synchronize(); t = time.perf_counter()
    cli.main(flags + ['--output_fname', os.path.join(d, f'out_w{w}.jsonl'), '--num_workers', str(w)])
    torch.cuda.synchronize(); out[w] = time.perf_counter() - t
    print(f'num_workers={w} host phases (ms per video, in call order): ' + '; '.join(f'{k}: ' + ' '.join(f'{x:.1f}' for x in v) for k, v in phase.items()), flush=True)
    phase.clear()
recs = {w: [json.loads(l) for l in open(os.path.join(d, f'out_w{w}.jsonl'))] for w in out}
same = all(recs[w] == recs[list(out)[0]] for w in out)
print(json.dumps(dict(clips=n_clips, frames_per_clip=frames, jpeg_frames_in_each_file=2 * frames, clip_bytes=os.path.getsize(os.path.join(d, 'clip0.avi')), make_clips_s=round(t_make, 1),
                      wall_s={f'num_workers={w}': round(v, 3) for w, v in out.items()}, frames_per_s={f'num_workers={w}': round(n_clips * frames / v, 1) for w, v in out.items()},
                      records_equal=same, records=len(recs[list(out)[0]]))), flush=True)
