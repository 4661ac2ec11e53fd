#!/usr/bin/env python
"""GEMM times of the LLM layer shapes at mid-size M (between the skinny and the chunk regime), auto dispatch."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'tools'))
import torch
from bench_gemm import run, LLM
from rawops import RawOps
ops = RawOps(torch.bfloat16)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for M in (49, 64, 65, 100, 128, 226, 256, 422, 512, 667, 800, 1000, 1274):
    tot = 0
    parts = []
    for name, N, K, epi in LLM[:4]:
        ms = run(ops, M, N, K, epi, variant, iters=10)
        tot += ms; parts.append(f'{name} {ms*1e3:7.1f}us')
    print(f'M={M:5d} layer {tot*1e3:8.1f} us  x28 = {tot*28:6.2f} ms | ' + ' '.join(parts), flush=True)
