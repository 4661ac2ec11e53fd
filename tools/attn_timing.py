#!/usr/bin/env python
"""Issue-time anatomy of the chunk attention loop (attn_gqa128_kernel<2>), DEBUG BUILD ONLY: `make -C mmduet_amd/csrc clean all ATTN_TIMING=1`, run this on the GPU box,
then rebuild plain (`make clean all`).  The stamps (s_memtime per segment, ~+10 % wave cycles) are summed over all waves of the launches between two resets."""
import ctypes as C, sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, bench
from mmduet_amd._lib import lib
sys.argv = [sys.argv[0]]
args = bench.parse(); args.multi_stream = 0
dev = torch.device('cuda', 0)
model, tok, cfg = bench.build(args, dev)
L = lib(); L.mmd_debug_attn_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; L.mmd_debug_attn_timing.restype = C.c_int
x = (torch.randn(1, 1274, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
cache = None; out = {}
for n_chunks in range(12):
    torch.cuda.synchronize(); L.mmd_debug_attn_timing(None, 1)
    o = model(inputs_embeds=x, past_key_values=cache); cache = o.past_key_values
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)(); L.mmd_debug_attn_timing(buf, 0)
    t = [int(v) for v in buf]
    if n_chunks in (0, 3, 7, 11) and t[4]:
        seg = t[0] + t[1] + t[2] + t[3]
        out[n_chunks * 1274] = dict(tiles_per_wave_total=t[4], cycles_per_wave_tile=round(seg / t[4]), wait_barrier_stage=round(t[0] / seg, 3), score_mfma=round(t[1] / seg, 3), softmax=round(t[2] / seg, 3),
                                    pv_mfma_or_dma_wait=round(t[3] / seg, 3), loop_share_of_kernel=round(seg / max(1, t[5]), 3))
        print(n_chunks * 1274, out[n_chunks * 1274], flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'attn_timing.json'), 'w'), indent=1)
