#!/usr/bin/env python
"""Run the LLM attention at one shape (for rocprofv3 --pmc): python tools/one_attn.py S n_ctx variant [iters]"""
import ctypes as C, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from mmduet_amd._lib import lib, check
from rawops import RawOps
S, n, v = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ops = RawOps(torch.bfloat16)
ms = C.c_float()
check(lib().mmd_op_attention_bench(ops.ctx, S, 28, 4, 128, n, v, iters, C.byref(ms)), ops.ctx)
print(f'S={S} n={n} v{v}: {ms.value*1e3:.1f} us  {4.0*S*(n+S)*128*28/ms.value/1e9:.1f} TF')
