#!/usr/bin/env python
"""Per-kernel averages of PMC counters from a rocprofv3 rocpd database: python tools/pmc_summary.py db [kernel-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ''
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print('columns:', cols, file=sys.stderr)
kn = 'kernel_name' if 'kernel_name' in cols else [c for c in cols if 'kernel' in c and 'name' in c][0]
cn = 'counter_name' if 'counter_name' in cols else [c for c in cols if 'counter' in c and 'name' in c][0]
vn = 'value' if 'value' in cols else [c for c in cols if 'value' in c][0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in cur.execute(f"select {kn}, {cn}, {vn} from counters_collection"):
    if sub in k:
        agg[k.split('(')[0][:70]][c].append(v)
for k, cs in agg.items():
    print(k)
    for c, vs in sorted(cs.items()):
        print(f'   {c:32s} n={len(vs):5d} avg={sum(vs)/len(vs):16.1f}')
