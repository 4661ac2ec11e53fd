"""Which rows / columns of the ViT ring attention differ from the register-staged kernel (diagnosis of a bad build): python tools/probes/d72_rows.py"""
import os, sys, torch, collections
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from rawops import RawOps
ops = RawOps(torch.bfloat16)
S, nh, d, n_ctx = 729, 16, 72, 0
g = torch.Generator().manual_seed(S * 7 + d)
cap = (n_ctx + S + 37 + 63) // 64 * 64
q = torch.randn(S, nh * d, generator=g); K = torch.randn(nh, cap, d, generator=g); V = torch.randn(nh, cap, d, generator=g)
Kd, Vd = K.to(ops.dev, ops.dtype), V.to(ops.dev, ops.dtype)
tag = os.environ.get('TAG', 'x')
o = ops.attention(q, Kd, Vd, nh, nh, d, n_ctx, False, 4).float().cpu().view(S, nh, d)
torch.save(o, f'/tmp/rows_{tag}.pt')
if os.environ.get('OTHER'):
    r = torch.load(f'/tmp/rows_{os.environ["OTHER"]}.pt')
    dd = (o - r).abs()
    bad_rows = (dd.amax((1, 2)) > 0).nonzero().flatten().tolist()
    print('rows differing', len(bad_rows), 'first', bad_rows[:40])
    print('row mod 32 histogram', sorted(collections.Counter(x % 32 for x in bad_rows).items()))
    print('row // 128 (block) histogram', sorted(collections.Counter(x // 128 for x in bad_rows).items()))
    bad_cols = (dd.amax((0, 1)) > 0).nonzero().flatten().tolist(); print('dims differing', bad_cols)
    bad_heads = (dd.amax((0, 2)) > 0).nonzero().flatten().tolist(); print('heads differing', bad_heads)
    print('max diff', dd.max().item())
    # is it the normalisation (a constant factor per row) or the sums themselves?
    for row in bad_rows[:3] + bad_rows[-2:]:
        ratio = (o[row, 0] / r[row, 0])
        print('row', row, 'head 0: ratio bad/good over dims: min %.4f max %.4f' % (ratio.min().item(), ratio.max().item()), ' good[:4]', r[row, 0, :4].tolist(), ' bad[:4]', o[row, 0, :4].tolist())
    o2 = ops.attention(q, Kd, Vd, nh, nh, d, n_ctx, False, 4).float().cpu().view(S, nh, d)
    print('repeat equal', torch.equal(o, o2), 'rows differing between repeats', int(((o - o2).abs().amax((1, 2)) > 0).sum()))
