#!/usr/bin/env python
"""Decode-side GEMVs at 1 <= M <= 32 rows (prompt / turn-prefix steps take the same kernels as a decode token): us per launch and the weight-stream rate.  python tools/probes/small_m_probe.py"""
import ctypes as C, sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from mmduet_amd._lib import lib, check, EPI
from rawops import RawOps
ops = RawOps(torch.bfloat16)
shapes = [('qkv', 4608, 3584, 'none'), ('o', 3584, 3584, 'none'), ('gate_up', 37888, 3584, 'swiglu'), ('down', 3584, 18944, 'none'), ('lm_head', 152064, 3584, 'none')]
for name, N, K, epi in shapes:
    W = (torch.randn(N, K, device=ops.dev) * 0.02).to(torch.bfloat16)
    for M in (1, 2, 4, 8, 12, 16, 17, 24, 32):
        X = (torch.randn(M, K, device=ops.dev) * 0.5).to(torch.bfloat16)
        ms = C.c_float(); plan = (C.c_int * 8)()
        check(lib().mmd_op_gemm_bench(ops.ctx, M, N, K, EPI[epi], 0, 20, C.byref(ms), C.c_void_p(X.data_ptr()), C.c_void_p(W.data_ptr())), ops.ctx)
        print(f'{name:8s} M={M:3d} {ms.value*1e3:8.1f} us  {N*K*2/ms.value/1e9:7.2f} TB/s', flush=True)
