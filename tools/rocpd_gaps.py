#!/usr/bin/env python
"""GPU idle-gap analysis of a rocprofv3 rocpd kernel trace: where (after which kernel / before which kernel) is the GPU idle?"""
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0   # us
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
tot_busy = 0; tot_gap = 0; last_end = rows[0][1]; last_name = ''
gaps = collections.Counter(); gapn = collections.Counter(); small = 0
for n, s, e in rows:
    n = re.sub(r'\(.*', '', n).replace('void ', '')[:40]
    g = (s - last_end) / 1e3
    if g > 0:
        tot_gap += g
        if g > thr:
            gaps[(last_name, n)] += g; gapn[(last_name, n)] += 1
        else:
            small += g
    tot_busy += (e - max(s, last_end)) / 1e3 if e > last_end else 0
    if e > last_end: last_end = e; last_name = n
print(f'span {(rows[-1][2]-rows[0][1])/1e6:.1f} ms  busy {tot_busy/1e3:.1f} ms  idle {tot_gap/1e3:.1f} ms (gaps<={thr}us: {small/1e3:.1f} ms over {len(rows)} launches)')
for k, v in gaps.most_common(18):
    print(f'  {v/1e3:8.1f} ms in {gapn[k]:5d} gaps (avg {v/gapn[k]:7.1f} us)  after {k[0]:40s} before {k[1]}')
