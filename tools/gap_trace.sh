#!/bin/bash
# GPU idle gaps of one timed-like stream pass (tower on its side stream, as in the timed region): rocprofv3 kernel trace -> tools/rocpd_gaps.py over the LAST stream pass
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_gap
rocprofv3 --kernel-trace -d $O/prof_gap -o trace -- python3 $R/bench.py --steps 1 --warmup 2 --no-prof --multi-stream 0 --no-cpu-baseline --no-parity-check --host-frames-steps 0 ${GAP_ARGS} > $O/gap_prof.log 2>&1
db=$(ls $O/prof_gap/*.db 2>/dev/null | head -1)
python3 - "$db" <<'PY' > $O/gap_report.txt
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
kt = [t for t in tabs if t.startswith('kernels')][0] if any(t.startswith('kernels') for t in tabs) else None
rows = cur.execute(f"select name, start, end from {kt} order by start").fetchall()
# the last stream pass = after the last big gap (> 5 ms: between steps the host rebuilds the driver)
segs = []; start = 0; run_end = rows[0][2]
for i in range(1, len(rows)):
    if rows[i][1] - run_end > 5e6: segs.append((start, i)); start = i
    run_end = max(run_end, rows[i][2])
segs.append((start, len(rows)))
# the timed-like pass = the LAST segment with more than 10000 launches (the model build, warm-up and the pass itself are separated by host pauses)
big = [sg for sg in segs if sg[1] - sg[0] > 10000]
a, b = big[-1] if big else max(segs, key=lambda sg: sg[1] - sg[0])
# the driver is reset without a host pause since round 5: the three passes (2 warm-up + 1 timed-like) then sit in ONE segment of 3 x the same launch count -- take the last third
if (b - a) > 60000: a = b - (b - a) // 3
rows = rows[a:b]
last_end = rows[0][1]; busy = 0; gaps = []; last = ''
for n, s, e in rows:
    n = re.sub(r'\(.*', '', n).replace('void ', '')[:44]
    if s > last_end: gaps.append(((s - last_end) / 1e3, last, n))
    if e > last_end:
        busy += (e - max(s, last_end)) / 1e3; last_end = e; last = n
span = (rows[-1][2] - rows[0][1]) / 1e3
print(f'last pass: {len(rows)} launches, span {span/1e3:.1f} ms, GPU busy (union over both streams) {busy/1e3:.1f} ms, idle {(span-busy)/1e3:.1f} ms = {(span-busy)/span*100:.1f} %')
for lo, hi in ((0, 5), (5, 20), (20, 100), (100, 1e9)):
    g = [x for x in gaps if lo <= x[0] < hi]
    print(f'  gaps {lo}-{hi} us: {len(g):6d}  total {sum(x[0] for x in g)/1e3:7.2f} ms')
c = collections.Counter(); cn = collections.Counter()
for g, a, b in gaps:
    if g >= 20: c[(a, b)] += g; cn[(a, b)] += 1
for k, v in c.most_common(12): print(f'   {v/1e3:7.2f} ms in {cn[k]:4d} gaps (avg {v/cn[k]:6.1f} us) after {k[0]} before {k[1]}')
PY
rm -rf $O/prof_gap
cat $O/gap_report.txt
