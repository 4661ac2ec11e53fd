#!/usr/bin/env python
"""Micro-benchmark the GEMM shapes of the path through mmd_op_gemm_bench (run on the GPU box), RANDOM operands
(operand bits set the chip's clock: zero / constant fills read 15-20 % high, MI355X guide rule 25).

    python tools/bench_gemm.py prod [out.json]     the production shapes (35-frame tower batch, 1274-row LLM chunk), dispatcher's choice
                                                   + every forced kernel: ms, TF/s, fraction of the 2.5 PF dense bf16 peak, kernel chosen
    python tools/bench_gemm.py llm                 weight-streaming shapes (M <= 64): GB/s
    python tools/bench_gemm.py big                 tile kernels over a range of M
"""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from mmduet_amd._lib import lib, check, EPI
from rawops import RawOps

LLM = [('qkv', 4608, 3584, 'none'), ('o', 3584, 3584, 'resid'), ('gate_up', 37888, 3584, 'swiglu'), ('down', 3584, 18944, 'resid'), ('lm_head', 152064, 3584, 'none')]
VIT = [('vit_patch', 1152, 640, 'none'), ('vit_qkv', 3456, 1152, 'none'), ('vit_o', 1152, 1152, 'resid'), ('vit_fc1', 4352, 1152, 'gelu_tanh'), ('vit_fc2', 1152, 4352, 'resid'),
       ('proj0', 3584, 1152, 'gelu_erf'), ('proj2', 3584, 3584, 'none')]
KERNELS = ('tile64', 'tile128', 'skinny', 'gemv16', 'big64', 'big128', 'ring256', 'ring128x2')
_cache = {}


def operands(ops, M, N, K):
    """Random bf16 operands at the magnitudes of the path (activations O(1), weights N(0, 0.02)), cached per shape."""
    key = (M, K)
    if key not in _cache:
        _cache.clear()
        _cache[key] = (torch.randn(M, K, device=ops.dev, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    X = _cache[key]
    W = (torch.randn(N, K, device=ops.dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    return X, W


def run(ops, M, N, K, epi, variant, iters=30, random=True):
    ms = C.c_float()
    X, W = operands(ops, M, N, K) if random else (None, None)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    check(lib().mmd_op_gemm_bench(ops.ctx, M, N, K, EPI[epi], variant, iters, C.byref(ms), p(X), p(W)), ops.ctx)
    plan = (C.c_int * 4)()
    lib().mmd_op_gemm_last_plan(ops.ctx, plan)
    run.plan = dict(kernel=KERNELS[plan[0]] if 0 <= plan[0] < len(KERNELS) else str(plan[0]), tiles=plan[1], splits=plan[2], blocks=plan[3])
    del W
    return ms.value


if __name__ == '__main__':
    ops = RawOps(torch.bfloat16)
    which = sys.argv[1] if len(sys.argv) > 1 else 'prod'
    if which == 'prod':
        rows = []
        variants = [(0, 'auto'), (4, 'big'), (6, 'ring256'), (16, 'rx-8w'), (17, 'rx-4w'), (24, 'rx-8w-ns4'), (32, 'rx-8w-early'), (33, 'rx-4w-early'), (40, 'rx-8w-ns4-early'), (42, 'rx-8w-m32-ns4-early'), (93, 'rx-fine'),
                    (7, 'ring256-splitK'), (44, 'rx-8w-ns4-early-splitK'), (49, 'rx-4w-nostagger'), (65, 'rx-4w-early-nostagger'), (96, 'DBG-x-contig'), (97, 'DBG-no-x'), (98, 'DBG-no-dma'), (99, 'DBG-no-epilogue'), (95, 'DBG-stores-pending'), (94, 'DBG-convert-no-store')]
        if len(sys.argv) > 3: variants = [v for v in variants if v[1] in sys.argv[3].split(',')]
        for M, shapes in ((25515, VIT), (1274, LLM[:4]), (1323, LLM[:4])):
            for name, N, K, epi in shapes:
                for variant, vn in variants:
                    if 'splitK' in vn and ((K < 8192 and not os.environ.get('BENCH_GEMM_SPLITK_ALL')) or epi == 'swiglu'):
                        continue
                    try:
                        ms = run(ops, M, N, K, epi, variant, iters=10)
                    except Exception as e:
                        continue
                    tf = 2 * M * N * K / ms / 1e9
                    NO = N // 2 if epi == 'swiglu' else N
                    alg = (M * K + N * K + M * NO + (M * NO if epi == 'resid' else 0)) * 2
                    rows.append(dict(M=M, name=name, N=N, K=K, epi=epi, variant=vn, ms=round(ms, 4), tflops=round(tf, 1), frac_of_2500=round(tf / 2500, 3),
                                     algorithmic_bytes=alg, **run.plan))
                    print(f'M={M:6d} {name:9s} N={N:6d} K={K:6d} {vn:16s} {ms*1e3:9.1f} us  {tf:7.1f} TF  {tf/2500:5.3f}  {run.plan}', flush=True)
        if len(sys.argv) > 2:
            json.dump(dict(note='tools/bench_gemm.py prod: mmd_op_gemm_bench, random bf16 operands (X ~ 0.5 N(0,1), W ~ 0.02 N(0,1)), 10 iterations after 3 warm-ups, '
                                'HIP events on the launch stream; frac = TF/s / 2500 (dense bf16 MFMA peak)', rows=rows), open(sys.argv[2], 'w'), indent=1)
    if which == 'ab':
        # interleaved A/B (guide rule 24): python tools/bench_gemm.py ab <variant,variant,...> [rounds] [out.json]; every round runs every variant on the same operands
        import statistics
        names = {200: 'ringw-3s3b', 201: 'ringw-4s2b', 202: 'ringw-4s3b', 204: 'ringw-3s3b-splitK', 0: 'auto', 4: 'big', 32: 'rx-8w-early', 33: 'rx-4w-early', 40: 'rx-8w-ns4-early', 92: 'DBG-ns4-half-barriers', 93: 'rx-fine', 99: 'DBG-no-epilogue', 98: 'DBG-no-dma', 7: 'ring-splitK', 48: 'rx-8w-early-splitK'}
        vs = [int(v) for v in sys.argv[2].split(',')]
        rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
        shapes = [(25515, 'vit_qkv', 3456, 1152, 'none'), (25515, 'vit_o_none', 1152, 1152, 'none'), (25515, 'vit_fc1', 4352, 1152, 'gelu_tanh'), (25515, 'vit_fc2', 1152, 4352, 'resid'), (25515, 'proj2', 3584, 3584, 'none'),
                  (1274, 'gate_up_none', 37888, 3584, 'none'), (1274, 'gate_up', 37888, 3584, 'swiglu'), (1274, 'down', 3584, 18944, 'resid'),
                  (1274, 'qkv', 4608, 3584, 'none'), (4096, 'sq4096', 4096, 4096, 'none'), (8192, 'sq8192', 8192, 8192, 'none')]
        if os.environ.get('AB_SHAPES'): shapes = [sh for sh in shapes if sh[1] in os.environ['AB_SHAPES'].split(',')]
        rows = []
        for M, name, N, K, epi in shapes:
            t = {v: [] for v in vs}
            for r in range(rounds):
                for v in vs:
                    try: t[v].append(run(ops, M, N, K, epi, v, iters=5))
                    except Exception as e: t[v].append(float('nan'))
            for v in vs:
                med, mn = statistics.median(t[v]), min(t[v])
                tf = 2 * M * N * K / med / 1e9
                rows.append(dict(M=M, name=name, N=N, K=K, variant=names.get(v, str(v)), median_us=round(med * 1e3, 1), min_us=round(mn * 1e3, 1), tflops_median=round(tf, 1), frac=round(tf / 2500, 3)))
                print(f'M={M:6d} {name:13s} N={N:6d} K={K:6d} {names.get(v, str(v)):18s} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  {tf:7.1f} TF  {tf/2500:5.3f}', flush=True)
        if len(sys.argv) > 4:
            json.dump(dict(note=f'tools/bench_gemm.py ab: {rounds} interleaved rounds x 5 iterations per variant, random bf16 operands, same operands for every variant of a shape', rows=rows), open(sys.argv[4], 'w'), indent=1)
    if which == 'lib':
        # reference point only (never on the product path): the ROCm library GEMM (torch.mm -> hipBLASLt / rocBLAS) on the same operands, interleaved with the dispatcher's kernel
        import statistics
        rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
        shapes = [(25515, n, N, K) for n, N, K, _ in VIT] + [(1274, n, N, K) for n, N, K, _ in LLM[:4]] + [(4096, 'sq4096', 4096, 4096), (8192, 'sq8192', 8192, 8192)]
        rows = []
        for M, name, N, K in shapes:
            X, W = operands(ops, M, N, K)
            Wt = W.t()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): torch.mm(X, Wt)
            t_lib, t_us = [], []
            for r in range(rounds):
                torch.cuda.synchronize(); e0.record()
                for _ in range(5): Y = torch.mm(X, Wt)
                e1.record(); torch.cuda.synchronize()
                t_lib.append(e0.elapsed_time(e1) / 5)
                t_us.append(run(ops, M, N, K, 'none', 0, iters=5))
            a, b = statistics.median(t_lib), statistics.median(t_us)
            rows.append(dict(M=M, name=name, N=N, K=K, library_us=round(a * 1e3, 1), ours_us=round(b * 1e3, 1), library_tflops=round(2 * M * N * K / a / 1e9, 1), ours_tflops=round(2 * M * N * K / b / 1e9, 1), ours_kernel=run.plan['kernel']))
            print(rows[-1], flush=True)
            del X, W, Wt
        if len(sys.argv) > 3:
            json.dump(dict(note=f'tools/bench_gemm.py lib: torch.mm (ROCm library GEMM, row-major X[M,K] x W[N,K]^T, bf16) beside mmd_op_gemm_bench variant 0 (epilogue none), {rounds} interleaved rounds x 5 iterations, random operands; '
                                'ours runs on different random operands of the same distribution (the bench entry draws its own W)', rows=rows), open(sys.argv[3], 'w'), indent=1)
    if which in ('llm', 'all'):
        rows = []
        for M in (1, 16, 49, 64):
            for name, N, K, epi in LLM:
                if name == 'lm_head' and M > 1: continue
                for variant, vn in ((2, 'skinny'), (5, 'skinny-slab'), (1002, 'fp8-skinny'), (1005, 'fp8-skinny-slab')):
                    if variant % 1000 == 5 and (epi == 'swiglu' or name == 'lm_head'): continue
                    ms = run(ops, M, N, K, epi, variant)
                    wb = 1 if variant >= 1000 else 2
                    gb = (N * K * wb + (M * K + M * (N // 2 if epi == 'swiglu' else N)) * 2) / 1e9
                    rows.append(dict(M=M, name=name, N=N, K=K, variant=vn, us=round(ms * 1e3, 2), GBps=round(gb / ms * 1e3), frac_of_8000=round(gb / ms * 1e3 / 8000, 3)))
                    print(f'M={M:4d} {name:8s} N={N:6d} K={K:6d} {vn:16s} {ms*1e3:8.1f} us  {gb/ms*1e3:7.0f} GB/s  {2*M*N*K/ms/1e9:7.1f} TF', flush=True)
        if len(sys.argv) > 2:
            json.dump(dict(note='tools/bench_gemm.py llm: weight-streaming kernels, random operands; GB/s = algorithmic bytes (weights at their stored width + activations) / time', rows=rows),
                      open(sys.argv[2], 'w'), indent=1)
    if which == 'stream':      # gemm_stream_kernel configuration sweep (DEBUG_VARIANTS build): python tools/bench_gemm.py stream [M]
        import statistics
        M = int(sys.argv[2]) if len(sys.argv) > 2 else 49
        cfgs = {2: 'skinny (shipped r03)', 300: 'WK2 KS2 NB4', 301: 'WK1 KS4 NB3', 302: 'WK4 KS1 NB4', 303: 'WK2 KS4 NB3', 304: 'WK1 KS4 NB4', 305: 'WK4 KS2 NB3', 307: 'WK2 KS2 NB6',
                320: 'WN5 / WN7x8 WK2 KS2 NB4', 321: 'WN5 WK3 / WN7x4', 322: 'DBG W only WN5 / WN7x8', 323: 'WN5 / WN7x8 WK1 KS4 NB3', 324: 'WN8 / WN7x16',
                310: 'DBG W only WK2', 311: 'DBG W only WK1', 312: 'DBG no W WK2', 313: 'DBG no W WK1'}
        for name, N, K, epi in LLM[:4]:
            t = {v: [] for v in cfgs}
            for r in range(5):
                for v in cfgs:
                    vv = v if v != 2 else (2 if epi == 'swiglu' else 5)
                    try: t[v].append(run(ops, M, N, K, epi, vv, iters=8))
                    except Exception as e: t[v].append(float('nan'))
            for v, nm in cfgs.items():
                med = statistics.median(t[v]); gb = N * K * 2 / 1e9
                print(f'STREAM M={M:4d} {name:8s} {nm:22s} median {med*1e3:8.1f} us   {gb/med*1e3:7.0f} GB/s of weights   {run.plan if v == 300 else ""}', flush=True)
    if which == 'square':       # the guide's reference shapes (4096^3, 8192^3): where does the ring stand against its 256^2 8-phase template (1.33 / 1.47 PF, random operands)?
        for n in (2048, 4096, 8192):
            for variant, vn in ((32, 'rx-8w-early'), (40, 'rx-8w-ns4-early'), (4, 'big')):
                ms = run(ops, n, n, n, 'none', variant, iters=10)
                print(f'{n}^3 {vn:16s} {ms*1e3:9.1f} us  {2*n**3/ms/1e9:7.1f} TF  {2*n**3/ms/1e9/2500:5.3f}', flush=True)
    if which in ('big', 'all'):
        for M in (392, 784, 980, 1274, 23328):
            shapes = (LLM[:4] if M < 2000 else VIT)
            for name, N, K, epi in shapes:
                for variant, vn in ((0, 'auto'), (4, 'big'), (6, 'ring256'), (7, 'ring256-splitK')):
                    ms = run(ops, M, N, K, epi, variant, iters=10)
                    print(f'M={M:6d} {name:8s} N={N:6d} K={K:6d} {vn:10s} {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:7.1f} TF', flush=True)
