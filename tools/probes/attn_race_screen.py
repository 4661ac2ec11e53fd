"""Race screen of the ring attention kernels: the same launch repeated many times must give the same bits every time (a DMA landing late or a slot refilled early shows as a
rare differing tile), also beside a second stream that keeps the memory system busy.  python tools/probes/attn_race_screen.py [repeats=300]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from rawops import RawOps
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ops = RawOps(torch.bfloat16)
dev = ops.dev
bad = 0
noise = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
for S, n_ctx in ((1274, 15000), (1274, 0), (637, 3000), (1911, 29000), (147, 1), (1, 15000), (2, 70001), (1, 200)):
    g = torch.Generator(device=dev).manual_seed(S + n_ctx)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, 28 * 128, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(4, cap, 128, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(4, cap, 128, generator=g, device=dev).to(torch.bfloat16)
    first = ops.attention(q, K, V, 28, 4, 128, n_ctx, True, 3).clone()
    diff = 0
    for r in range(reps):
        if r % 3 == 0:
            with torch.cuda.stream(side):          # a concurrent copy stream: perturbs DMA arrival times
                noise[:256 << 20].copy_(noise[256 << 20:], non_blocking=True)
        o = ops.attention(q, K, V, 28, 4, 128, n_ctx, True, 3)
        if not torch.equal(o, first): diff += 1
    torch.cuda.synchronize()
    print(f'S={S} n_ctx={n_ctx}: {diff} of {reps} repeats differ from the first', flush=True)
    bad += diff
# the ViT ring kernel (attn_d72_ring_kernel: head_dim 72, bidirectional; the raw op runs one sequence, so many heads stand in for the batch: 6 x 560 = 3 360 blocks as in a 35-frame batch)
for S, nh, n_ctx in ((729, 560, 0), (196, 560, 533), (729, 16, 0), (300, 64, 0)):
    g = torch.Generator(device=dev).manual_seed(S + nh)
    cap = (n_ctx + S + 100 + 63) // 64 * 64
    q = torch.randn(S, nh * 72, generator=g, device=dev).to(torch.bfloat16)
    K = torch.randn(nh, cap, 72, generator=g, device=dev).to(torch.bfloat16)
    V = torch.randn(nh, cap, 72, generator=g, device=dev).to(torch.bfloat16)
    first = ops.attention(q, K, V, nh, nh, 72, n_ctx, False, 4).clone()
    diff = 0
    for r in range(reps):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                noise[:256 << 20].copy_(noise[256 << 20:], non_blocking=True)
        if not torch.equal(ops.attention(q, K, V, nh, nh, 72, n_ctx, False, 4), first): diff += 1
    torch.cuda.synchronize()
    print(f'ViT ring S={S} heads={nh} n_ctx={n_ctx}: {diff} of {reps} repeats differ from the first', flush=True)
    bad += diff
print('RACE SCREEN', 'CLEAN' if bad == 0 else f'FAILED ({bad})')
sys.exit(1 if bad else 0)
