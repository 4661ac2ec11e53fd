#!/usr/bin/env python
"""Which kernels run INSIDE a stream?  From a rocprofv3 rocpd kernel trace of `bench.py --steps 1 --warmup 0 ...`: the window from the first to the last LLM step of
the LAST stream pass in the trace (its first / last gemm_ringx launch with the SwiGLU epilogue = gate_up of a chunk forward), and every kernel name in it with counts.
Evidence for "no at::native / torch kernel between two forwards of a stream" (VERDICT r02 item 6)."""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
gate = [i for i, r in enumerate(rows) if 'gemm_ringx_kernel<4' in r[0] or 'gemm_ringx_kernelILi4' in r[0]]
if not gate:
    print('no chunk forward (SwiGLU ring GEMM) in the trace'); sys.exit(0)
# stream passes are separated by long gaps without gate_up launches (weight set-up, the CPU leg): take the last run of launches
runs, cur_run = [], [gate[0]]
for a, b in zip(gate, gate[1:]):
    if rows[b][1] - rows[a][2] > 400e6: runs.append(cur_run); cur_run = []
    cur_run.append(b)
runs.append(cur_run)
first, last = runs[-1][0], runs[-1][-1]
cnt = collections.Counter(re.sub(r'\(.*', '', r[0]).replace('void ', '')[:100] for r in rows[first:last + 1])
span = (rows[last][2] - rows[first][1]) / 1e6
print(f'window: kernels {first}..{last} of {len(rows)} ({last - first + 1} dispatches, {span:.1f} ms) = first to last chunk forward of the last stream pass in the trace')
foreign = {k: v for k, v in cnt.items() if k.startswith('at::') or 'at::native' in k or 'elementwise_kernel' in k}
print('torch (at::native) kernels in the window:', foreign if foreign else 'NONE')
print('runtime blit kernels in the window:', {k: v for k, v in cnt.items() if k.startswith('__amd_rocclr')})
for k, v in cnt.most_common():
    print(f'{v:7d}  {k}')
