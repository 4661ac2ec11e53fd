import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_gemm import run
from rawops import RawOps
ops = RawOps(torch.bfloat16)
for (M, N, K) in [(8192, 8192, 8192), (4096, 4096, 16384), (8192, 8192, 1152), (8192, 8192, 256), (23328, 4352, 1152), (23296, 4352, 1152), (23328, 4096, 1152)]:
    for v, vn in ((4, 'big'), (6, 'ring256')):
        ms = run(ops, M, N, K, 'none', v, iters=10)
        print(f'M={M} N={N} K={K} {vn:7s} {ms*1e3:9.1f} us {2*M*N*K/ms/1e9:7.1f} TF', flush=True)
