#!/usr/bin/env python
"""Ordering probe of the graph-replayed tower batches: fa / fb alternate through the staging buffers (fb's results dropped at once, so the allocator hands the same block to the
next call); which result does every fa call return?   tower_graph_probe.py [B] [side]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_production import _build
m, w, o = _build(2, 2, torch.bfloat16)
g = torch.Generator(device=m.device).manual_seed(21)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 35
side = torch.cuda.Stream() if len(sys.argv) > 2 else None
fa = torch.randint(0, 256, (B, 3, 336, 336), generator=g, device=m.device, dtype=torch.uint8)
fb = torch.randint(0, 256, (B, 3, 336, 336), generator=g, device=m.device, dtype=torch.uint8)
torch.cuda.synchronize()
def run():
    ra = m.visual_embed_frames(fa).clone(); torch.cuda.synchronize()
    outs = []
    for i in range(10):
        m.visual_embed_frames(fb)
        outs.append(m.visual_embed_frames(fa).clone())
    torch.cuda.synchronize()
    return [('A' if torch.equal(t, ra) else ('~' if torch.equal(t[-49:], ra[-49:]) else '?')) for t in outs]
if side is not None:
    with torch.cuda.stream(side):
        print('side stream:', run())
else:
    print('default stream:', run())
