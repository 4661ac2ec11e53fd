#!/usr/bin/env python
"""Vision tower time per frame against the batch size: 256x256 tiles over 256 CUs quantise, the best batch is not a power of two."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd.configuration_live import VideoHeadLiveLlavaQwenConfig
from mmduet_amd.modeling_live import VideoHeadLiveLlavaQwenForCausalLM
from mmduet_amd.weights import synthetic_weights
dev = torch.device('cuda', 0)
cfg = VideoHeadLiveLlavaQwenConfig(frame_num_tokens=49, frame_resolution=384, v_placeholder='<image>', num_hidden_layers=1)
model = VideoHeadLiveLlavaQwenForCausalLM(cfg, torch_dtype=torch.bfloat16, device=dev, max_vit_batch=48, max_step_tokens=256, kv_initial_tokens=1024)
for name, t in synthetic_weights(cfg, seed=0, device=dev, dtype=torch.bfloat16, scale='init02'):
    model.load_tensor(name, t)
model.finalize()
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '24,28,30,31,32,33,34,35,36,40,44,48').split(',')]:
    px = torch.randn(B, 3, 384, 384, device=dev).to(torch.bfloat16)
    model.visual_embed(px); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): model.visual_embed(px)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'B={B:3d}: {dt*1e3:7.2f} ms per batch, {dt*1e6/B:7.1f} us per frame', flush=True)
