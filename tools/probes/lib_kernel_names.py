"""Which library kernels does torch.mm pick for the path's shapes?  (reference point only; run under rocprofv3 --kernel-trace --stats)"""
import torch
shapes = [(25515, 3456, 1152), (25515, 1152, 4352), (25515, 3584, 3584), (1274, 4608, 3584), (1274, 37888, 3584), (4096, 4096, 4096), (8192, 8192, 8192)]
for M, N, K in shapes:
    X = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
    for _ in range(4):
        torch.mm(X, W.t())
    torch.cuda.synchronize()
