#!/usr/bin/env python
"""tower_graph_probe2.py [sync]: the failing sequence of test_graph_replayed_tower_batches_give_the_launches_bits, with the frames whose rows differ listed."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_production import _build
sync = len(sys.argv) > 1 and sys.argv[1] == 'sync'
pre = len(sys.argv) > 2
m, w, o = _build(2, 2, torch.bfloat16)
g = torch.Generator(device=m.device).manual_seed(21)
B = 35
fa = torch.randint(0, 256, (B, 3, 336, 336), generator=g, device=m.device, dtype=torch.uint8)
fb = torch.randint(0, 256, (B, 3, 336, 336), generator=g, device=m.device, dtype=torch.uint8)
if pre:
    px = m.get_vision_tower().image_processor.preprocess(fa, return_tensors='pt')['pixel_values'].to(torch.bfloat16)
    ref = m.visual_embed(px)
outs = []
for rep in range(5):
    outs.append(m.visual_embed_frames(fa).clone())
    if sync: torch.cuda.synchronize()
    m.visual_embed_frames(fb)
    if sync: torch.cuda.synchronize()
torch.cuda.synchronize()
for i, t in enumerate(outs):
    bad = [f for f in range(B) if not torch.equal(t[f * 49:(f + 1) * 49], outs[0][f * 49:(f + 1) * 49])]
    print(i, 'frames that differ from the first (launched) result:', bad)
