#!/usr/bin/env python
"""Is torch's float8_e4m3fn cast ON THE GPU the same function as on the CPU (the quantiser is pinned bit-exact against the CPU cast)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
g = torch.Generator().manual_seed(0)
W = (torch.randn(4608, 3584, generator=g) * 0.02).to(torch.bfloat16)
def deq(W):
    vf = W.float(); amax = vf.abs().amax(dim=1); scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    a = vf / scale[:, None]
    return a, a.to(torch.float8_e4m3fn), scale
a_c, q_c, s_c = deq(W)
a_g, q_g, s_g = deq(W.cuda())
print('CAST scale equal', torch.equal(s_c, s_g.cpu()), ' ratio equal', torch.equal(a_c, a_g.cpu()))
qc, qg = q_c.view(torch.uint8), q_g.cpu().view(torch.uint8)
ne = (qc != qg)
print(f'CAST mismatching fp8 codes: {int(ne.sum())} of {ne.numel()} ({100.0 * ne.float().mean().item():.4f} %)')
if ne.any():
    idx = ne.nonzero()[:8]
    for i, j in idx.tolist():
        print('CAST   value', float(a_c[i, j]), 'cpu', float(q_c[i, j].float()), 'gpu', float(q_g.cpu()[i, j].float()))
    # same INPUT values cast on both devices
    q2 = a_c.cuda().to(torch.float8_e4m3fn).cpu().view(torch.uint8)
    print('CAST same fp32 inputs, cast on gpu vs cpu: mismatches', int((q2 != qc).sum()))

from oracle.stream_check import quantise_e4m3_rows
q_d, s_d = quantise_e4m3_rows(W.cuda())
print('CAST double-division form on the GPU vs the CPU cast: scale equal', torch.equal(s_d.cpu(), s_c), ' code mismatches', int((q_d.cpu().view(torch.uint8) != qc).sum()))
q_dc, s_dc = quantise_e4m3_rows(W)
print('CAST double-division form on the CPU vs the plain CPU cast: scale equal', torch.equal(s_dc, s_c), ' code mismatches', int((q_dc.view(torch.uint8) != qc).sum()))
