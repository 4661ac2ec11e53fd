#!/usr/bin/env python
"""Issue-time anatomy of the SMALL-S split-KV attention (S <= 64: per-frame steps and decode), DEBUG LIBRARY ONLY:
   hipcc ... -DMMDUET_ATTN_TIMING -c attn.hip -o /tmp/attn_timing.o; link as mmduet_amd/csrc/libmmduet_hip_timing.so (never shipped, git-ignored).
   python tools/attn_small_timing.py [S n_ctx ...]   -> per-wave cycles of the loop segments, summed over all waves of the launches between two resets."""
import ctypes as C, sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import mmduet_amd._lib as L_
L_.LIB_PATH = os.path.join(R, 'mmduet_amd', 'csrc', 'libmmduet_hip_timing.so')
import torch
from mmduet_amd._lib import lib, check
from rawops import RawOps
ops = RawOps(torch.bfloat16)
L = lib(); L.mmd_debug_attn_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; L.mmd_debug_attn_timing.restype = C.c_int
shapes = [(49, 1024), (49, 4096), (49, 15000), (49, 30000), (24, 15000), (64, 15000)]
if len(sys.argv) > 2: shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
out = {}
iters = 20
for S, n in shapes:
    ms = C.c_float()
    check(L.mmd_op_attention_bench(ops.ctx, S, 28, 4, 128, n, 3, 3, C.byref(ms)), ops.ctx)          # warm
    torch.cuda.synchronize(); L.mmd_debug_attn_timing(None, 1)
    check(L.mmd_op_attention_bench(ops.ctx, S, 28, 4, 128, n, 3, iters, C.byref(ms)), ops.ctx)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)(); L.mmd_debug_attn_timing(buf, 0)
    t = [int(v) for v in buf]
    launches = iters + 3
    seg = t[0] + t[1] + t[2] + t[3]
    rec = dict(us_instrumented=round(ms.value * 1e3, 2), wave_tiles_per_launch=t[4] / launches, cycles_per_wave_tile=round(seg / max(1, t[4])),
               wait_barrier_stage=round(t[0] / max(1, seg), 3), score_mfma=round(t[1] / max(1, seg), 3), softmax=round(t[2] / max(1, seg), 3), pv_mfma=round(t[3] / max(1, seg), 3),
               loop_share_of_kernel=round(seg / max(1, t[5]), 3), kernel_cycles_per_wave_sum_per_launch=round(t[5] / launches))
    out[f'S={S} n={n}'] = rec
    print(f'S={S} n={n}', rec, flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'r05_attn_small_timing.json'), 'w'), indent=1)
