"""ViT attention, register-staged kernel vs the DMA ring (MMDUET_VIT_ATTN_RING = 0 / 3 / 2, read once per process): digest of the tower output on seeded frames
(the kernels do the same arithmetic in the same order: the digests must agree) and the per-layer time inside the model.   python tools/probes/vit_ring_ab.py"""
import sys, os, json, hashlib, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch, bench
    sys.argv = [sys.argv[0]]
    args = bench.parse(); args.multi_stream = 0
    dev = torch.device('cuda', 0)
    model, tok, cfg = bench.build(args, dev)
    g = torch.Generator().manual_seed(3)
    out = {}
    for nf in (35, 3):
        frames = torch.randint(0, 256, (nf, 3, 336, 336), generator=g, dtype=torch.uint8).to(dev)
        y = model.visual_embed_frames(frames); torch.cuda.synchronize()
        out[f'pooled_{nf}'] = hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16]
        model.set_full_tower(True)
        y = model.visual_embed_frames(frames); torch.cuda.synchronize()
        out[f'pooled_full_tower_{nf}'] = hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16]
        model.set_full_tower(False)
    px = torch.randn(35, 3, 384, 384, device=dev).to(torch.bfloat16)
    ts = []
    for r in range(5):
        model.prof_reset(); model.prof_set_stride(1); model.prof_enable(['attn_vit'])
        model.visual_embed(px); torch.cuda.synchronize(); model.prof_enable(False)
        p = model.prof_read()['attn_vit']; ts.append(p['ms'] / p['launches'] * 1e3)
    out['us_per_layer'] = round(sorted(ts)[2], 1)
    print('RES ' + json.dumps(out))
    sys.exit(0)
res = {}
for mode in (os.environ.get('RING_MODES', '0,1,0,1').split(',')):
    extra = dict(MMDUET_VIT_ATTN_RING=mode)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith('RES ')]
    if not line:
        print('mode', mode, 'FAILED', r.stderr[-1500:]); continue
    d = json.loads(line[0][4:]); print('mode', mode, d, flush=True)
    res.setdefault(mode, []).append(d)
ok = all(all(d[k] == res['0'][0][k] for k in d if k != 'us_per_layer') for m in res for d in res[m])
print('digests equal across modes:', ok)
