#!/usr/bin/env python
"""RMS error of the fp8-weight GEMM regimes against fp32 math on the dequantised weights, next to the plain bf16 GEMM's -- is the extra mean head-logit error of the fp8
build (tools/probes/fp8_mean_probe.py: B) inside the GEMMs?      python tools/probes/fp8_gemm_rms_probe.py"""
import os, sys, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.nn.functional as F
from rawops import RawOps
import test_gpu_fp8 as T8
ops = RawOps(torch.bfloat16)
dev = ops.dev
for (N, K, epi) in [(4608, 3584, 'none'), (3584, 3584, 'resid'), (37888, 3584, 'swiglu'), (3584, 18944, 'resid')]:
    for M in (1, 49, 1911):
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        X = (torch.randn(M, K, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
        Rr = (torch.randn(M, N, generator=g, device=dev)).to(torch.bfloat16) if epi == 'resid' else None
        def ref_of(Wd):
            lin = F.linear(X.double(), Wd.double())
            if epi == 'resid': return lin, lin.float().to(torch.bfloat16).double() + Rr.double()
            if epi == 'swiglu':
                v = lin.view(M, -1, 2, 16); gg, uu = v[:, :, 0].reshape(M, -1), v[:, :, 1].reshape(M, -1)
                return lin, F.silu(gg) * uu
            return lin, lin
        Wq, q8, sc = T8.hip_quantize(ops, W)
        Wd = Wq.float() * sc[:, None]
        Y8 = T8.hip_gemm_w8(ops, X, Wq, q8, sc, None, Rr, epi).double()
        _, ref8 = ref_of(Wd)
        Y16 = ops.gemm(X, W, None, R=Rr, epi=epi, variant=0).double()
        _, ref16 = ref_of(W)
        if epi == 'resid':          # the rounding of interest is the linear's: take the residual back out
            Y8, ref8, Y16, ref16 = Y8 - Rr.double(), ref8 - Rr.double(), Y16 - Rr.double(), ref16 - Rr.double()
        r8 = ((Y8 - ref8).pow(2).mean().sqrt() / ref8.pow(2).mean().sqrt()).item()
        r16 = ((Y16 - ref16).pow(2).mean().sqrt() / ref16.pow(2).mean().sqrt()).item()
        print(f'RMS N={N} K={K} {epi:7s} M={M:5d}: fp8 path {r8:.3e}   bf16 path {r16:.3e}   ratio {r8 / r16:.3f}', flush=True)
