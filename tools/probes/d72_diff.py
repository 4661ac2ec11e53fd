import os, sys, torch
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from rawops import RawOps
ops = RawOps(torch.bfloat16)
for (S, nh, d, n_ctx) in [(729, 16, 72, 0), (70, 2, 72, 58), (196, 16, 72, 533)]:
    g = torch.Generator().manual_seed(S * 7 + d)
    cap = (n_ctx + S + 37 + 63) // 64 * 64
    q = torch.randn(S, nh * d, generator=g); K = torch.randn(nh, cap, d, generator=g); V = torch.randn(nh, cap, d, generator=g)
    Kd, Vd = K.to(ops.dev, ops.dtype), V.to(ops.dev, ops.dtype)
    o1 = ops.attention(q, Kd, Vd, nh, nh, d, n_ctx, False, 4).float().clone()
    o2 = ops.attention(q, Kd, Vd, nh, nh, d, n_ctx, False, 4).float().clone()
    torch.save(o1.cpu(), f'/tmp/o_{os.environ.get("TAG","x")}_{S}.pt')
    print(S, nh, n_ctx, 'repeat equal', torch.equal(o1, o2), 'absmax', o1.abs().max().item())
    other = f'/tmp/o_{os.environ.get("OTHER","")}_{S}.pt'
    if os.path.exists(other):
        o0 = torch.load(other).to(o1.device)
        dd = (o1 - o0).abs(); print('   vs', os.environ['OTHER'], 'max diff', dd.max().item(), 'n diff', int((dd > 0).sum()), 'of', dd.numel(), 'rows differing', int((dd.amax(1) > 0).sum()))
        kk = V[:, n_ctx + S:, :]
        print('   V past the end: absmax', kk.abs().max().item())
