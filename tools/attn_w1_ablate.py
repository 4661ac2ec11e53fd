#!/usr/bin/env python
"""GPU half of tools/attn_w1_ablate.sh: wall of variant 5 at the chunk / per-frame shapes through one library: python tools/attn_w1_ablate.py <lib.so> <tag>"""
import ctypes as C, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import mmduet_amd._lib as L_
L_.LIB_PATH = os.path.join(R, 'mmduet_amd', 'csrc', sys.argv[1])
import torch
from mmduet_amd._lib import lib, check
from rawops import RawOps
ops = RawOps(torch.bfloat16)
out = []
for S, n, v in ((1274, 15000, 5), (1274, 15000, 3), (49, 15000, 5), (392, 15000, 5)):
    ms = C.c_float()
    check(lib().mmd_op_attention_bench(ops.ctx, S, 28, 4, 128, n, v, 20, C.byref(ms)), ops.ctx)
    out.append(f'S={S} n={n} v{v}: {ms.value * 1e3:7.1f} us')
print(f'{sys.argv[2]:28s}', ' | '.join(out), flush=True)
