#!/usr/bin/env python
"""Micro-benchmark the LLM attention through mmd_op_attention_bench (run on the GPU box)."""
import ctypes as C, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from mmduet_amd._lib import lib, check
from rawops import RawOps
ops = RawOps(torch.bfloat16)
small = len(sys.argv) > 1 and sys.argv[1] == 'small'          # the split-KV forms only (S <= 64), shipped kernel only
for S in ((1, 4, 24, 49, 64) if small else (1, 49, 392, 1274)):
    for n in (0, 1024, 4096, 15000, 30000):
        for v in ((3,) if small else (2, 3)):
            ms = C.c_float()
            check(lib().mmd_op_attention_bench(ops.ctx, S, 28, 4, 128, n, v, 20, C.byref(ms)), ops.ctx)
            fl = 4.0 * S * (n + S) * 128 * 28
            kvb = 2.0 * (n + S) * 4 * 128 * 2
            print(f'S={S:4d} n={n:6d} variant={v} {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:8.1f} TF  KV {kvb/ms.value/1e6:7.0f} GB/s', flush=True)
