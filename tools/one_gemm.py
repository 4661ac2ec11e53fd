#!/usr/bin/env python
"""Run one GEMM shape a few times (for rocprofv3 --pmc passes): python tools/one_gemm.py M N K epi variant [iters]"""
import ctypes as C, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from mmduet_amd._lib import lib, check, EPI
from rawops import RawOps
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); epi = sys.argv[4]; variant = int(sys.argv[5]); iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
ops = RawOps(torch.bfloat16, max_step_tokens=4096 if M > 2048 else 64)          # (merged chunks of several streams: the large slab workspace, as in the model)
ms = C.c_float()
X = (torch.randn(M, K, device=ops.dev) * 0.5).to(torch.bfloat16); W = (torch.randn(N, K, device=ops.dev) * 0.02).to(torch.bfloat16)          # random operands: operand bits set the clock
if os.environ.get('ONE_GEMM_CONSTANT'):          # a constant fill instead (mmd_op_gemm_bench's own): how far the clock moves with the operand bits
    check(lib().mmd_op_gemm_bench(ops.ctx, M, N, K, EPI[epi], variant, iters, C.byref(ms), None, None), ops.ctx)
else:
    check(lib().mmd_op_gemm_bench(ops.ctx, M, N, K, EPI[epi], variant, iters, C.byref(ms), C.c_void_p(X.data_ptr()), C.c_void_p(W.data_ptr())), ops.ctx)
print(f'{M}x{N}x{K} {epi} v{variant}: {ms.value*1e3:.1f} us {2*M*N*K/ms.value/1e9:.1f} TF')
