"""LLM chunk qkv / o / (gate_up, down for reference) at the chunk sizes of the configs: dispatcher's choice against the 4-wave 256x128 ring (variant 33) and the 128-row kernel."""
import sys, os, json, statistics
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench_gemm as B
from rawops import RawOps
ops = RawOps(torch.bfloat16)
rows = []
for M in (245, 392, 637, 784, 980, 1274, 1323, 1911, 1960, 2548):
    for name, N, K, epi in B.LLM[:4]:
        t = {}
        for v in (0, 4, 33):
            if v == 33 and epi == 'swiglu' and False: continue
            ts = []
            for r in range(5):
                try: ts.append(B.run(ops, M, N, K, epi, v, iters=5))
                except Exception as e: ts.append(float('nan'))
            t[v] = statistics.median(ts)
            if v == 0: k0 = B.run.plan['kernel']
        rows.append(dict(M=M, name=name, auto_us=round(t[0] * 1e3, 1), auto_kernel=k0, big_us=round(t[4] * 1e3, 1), ring4w_us=round(t[33] * 1e3, 1)))
        print(rows[-1], flush=True)
json.dump(rows, open(os.path.join(R, 'gpurun_out', 'midm_ring4w_sweep.json'), 'w'), indent=1)
