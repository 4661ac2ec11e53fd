#!/usr/bin/env python
"""The vendor library's GEMM (torch.matmul -> hipBLASLt / rocBLAS) on one shape, random or constant bf16 operands, for a rocprofv3 --pmc pass:
python tools/lib_gemm_clock.py M N K [constant]   (prints wall per call from HIP events)"""
import sys, torch
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); const = len(sys.argv) > 4
dev = torch.device('cuda', 0)
if const:
    X = torch.full((M, K), 0.5, device=dev, dtype=torch.bfloat16); W = torch.full((N, K), 0.0195, device=dev, dtype=torch.bfloat16)
else:
    X = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
for _ in range(3): Y = X @ W.T
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8): Y = X @ W.T
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 8
print(f'{M}x{N}x{K} {"constant" if const else "random"}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.1f} TF')
