"""Dispatcher rule check: ~one block wave of 256^2 tiles (231..256) -- ring (auto) against the 128-row kernel (variant 4), short and long K."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, statistics
import bench_gemm as B
from rawops import RawOps
ops = RawOps(torch.bfloat16)
for M, N, K in ((3840, 4096, 1152), (3840, 4096, 3584), (4096, 4096, 4096), (4096, 3840, 1152), (3072, 5120, 3584)):
    t = {0: [], 4: []}
    for r in range(5):
        for v in (0, 4):
            t[v].append(B.run(ops, M, N, K, 'none', v, iters=5)); 
            if v == 0: k0 = B.run.plan['kernel']
    a, b = statistics.median(t[0]), statistics.median(t[4])
    print(f'M={M} N={N} K={K}: auto ({k0}) {a*1e3:.1f} us, 128-row {b*1e3:.1f} us', flush=True)
