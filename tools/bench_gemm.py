#!/usr/bin/env python
"""Micro-benchmark the GEMM shapes of the path through mmd_op_gemm_bench (run on the GPU box)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from mmduet_amd._lib import lib, check, EPI
from rawops import RawOps

LLM = [('qkv', 4608, 3584, 'none'), ('o', 3584, 3584, 'resid'), ('gate_up', 37888, 3584, 'swiglu'), ('down', 3584, 18944, 'resid'), ('lm_head', 152064, 3584, 'none')]
VIT = [('vit_qkv', 3456, 1152, 'none'), ('vit_o', 1152, 1152, 'resid'), ('vit_fc1', 4352, 1152, 'gelu_tanh'), ('vit_fc2', 1152, 4352, 'resid'), ('proj0', 3584, 1152, 'gelu_erf'), ('proj2', 3584, 3584, 'none')]

def run(ops, M, N, K, epi, variant, iters=30):
    ms = C.c_float()
    check(lib().mmd_op_gemm_bench(ops.ctx, M, N, K, EPI[epi], variant, iters, C.byref(ms)), ops.ctx)
    return ms.value

if __name__ == '__main__':
    ops = RawOps(torch.bfloat16)
    which = sys.argv[1] if len(sys.argv) > 1 else 'llm'
    if which in ('llm', 'all'):
        for M in (1, 16, 49, 64):
            for name, N, K, epi in LLM:
                if name == 'lm_head' and M > 1: continue
                for variant, vn in ((2, 'skinny'), (5, 'skinny-slab')):
                    if variant == 5 and (epi == 'swiglu' or name == 'lm_head'): continue
                    ms = run(ops, M, N, K, epi, variant)
                    gb = (N * K + M * K + M * (N // 2 if epi == 'swiglu' else N)) * 2 / 1e9
                    print(f'M={M:4d} {name:8s} N={N:6d} K={K:6d} {vn:10s} {ms*1e3:8.1f} us  {gb/ms*1e3:7.0f} GB/s  {2*M*N*K/ms/1e9:7.1f} TF', flush=True)
    if which in ('big', 'all'):
        for M in (392, 784, 980, 1274, 23328):
            shapes = (LLM[:4] if M < 2000 else VIT)
            for name, N, K, epi in shapes:
                for variant, vn in ((0, 'auto'), (4, 'big'), (6, 'ring256'), (7, 'ring256-splitK')):
                    ms = run(ops, M, N, K, epi, variant, iters=10)
                    print(f'M={M:6d} {name:8s} N={N:6d} K={K:6d} {vn:10s} {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:7.1f} TF', flush=True)
