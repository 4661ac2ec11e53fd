#!/bin/bash
# GPU idle gaps of the multi-stream response leg: rocprofv3 kernel trace of tools/multistream_anatomy.py, gap analysis over the window of its LAST (timed) pass
# usage: tools/gap_trace_multi.sh [4x13] [videos per slot]   -> gpurun_out/gap_multi_report.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
CFG=${1:-4x13}; PER=${2:-1}; PH=${3:-ab}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_gapm
rocprofv3 --kernel-trace -d $O/prof_gapm -o trace -- python3 $R/tools/multistream_anatomy.py $CFG $PER $PH > $O/gap_multi_prof.log 2>&1
db=$(ls $O/prof_gapm/*.db 2>/dev/null | head -1)
python3 - "$db" "$O/multistream_anatomy.json" "$CFG" <<'PY' > $O/gap_multi_report.txt
import sqlite3, sys, re, collections, json
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
wall_ms = json.load(open(sys.argv[2]))[sys.argv[3]]['wall_ms']
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
kt = [t for t in tabs if t.startswith('kernels')][0]
rows = cur.execute(f"select name, start, end from {kt} order by start").fetchall()
t_end = max(r[2] for r in rows); t_beg = t_end - wall_ms * 1e6
rows = [r for r in rows if r[1] >= t_beg]
def short(n): return re.sub(r'\(.*', '', n).replace('void ', '')[:60]
last_end = rows[0][1]; busy = 0; gaps = []; last = ''
tower = 0.0
for n, s, e in rows:
    n = short(n)
    if s > last_end: gaps.append(((s - last_end) / 1e3, last, n))
    if e > last_end:
        busy += (e - max(s, last_end)) / 1e3; last_end = e; last = n
span = (rows[-1][2] - rows[0][1]) / 1e3
print(f'{sys.argv[3]} timed pass (wall {wall_ms:.0f} ms by the host clock): {len(rows)} launches, span {span/1e3:.1f} ms, GPU busy (union over the streams) {busy/1e3:.1f} ms, idle {(span-busy)/1e3:.1f} ms = {(span-busy)/span*100:.1f} %')
for lo, hi in ((0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    g = [x for x in gaps if lo <= x[0] < hi]
    print(f'  gaps {lo}-{hi} us: {len(g):6d}  total {sum(x[0] for x in g)/1e3:7.2f} ms')
c = collections.Counter(); cn = collections.Counter()
for g, a, b in gaps:
    if g >= 20: c[(a, b)] += g; cn[(a, b)] += 1
for k, v in c.most_common(16): print(f'   {v/1e3:7.2f} ms in {cn[k]:4d} gaps (avg {v/cn[k]:6.1f} us) after {k[0]} before {k[1]}')
ks = collections.Counter(); kn = collections.Counter()
for n, s, e in rows: ks[short(n)] += (e - s) / 1e6; kn[short(n)] += 1
print('kernel time in the window (ms, launches):')
for k, v in ks.most_common(45): print(f'   {v:8.1f} {kn[k]:6d}  {k}')
PY
rm -rf $O/prof_gapm
cat $O/gap_multi_report.txt
