#!/bin/bash
# CLI clip prefetch: A/B wall (no profiler), then the prefetch run alone under a kernel trace -> GPU idle gaps between videos.  -> gpurun_out/<tag>_cli_prefetch.txt
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CLI_PROBE_DIR=/tmp/mmduet_cli_probe
{
echo "# python tools/cli_prefetch_probe.py 4 8 120: the CLI over 8 Motion-JPEG clips (240 JPEG frames each, 120 kept at 1 fps), 7B + so400m, k = 26, no responses; num_workers 0 = inline loader (round 4)"
python3 $R/tools/cli_prefetch_probe.py 4 8 120 2>/dev/null | tail -3
rm -rf $O/prof_cli
CLI_PROBE_AB=0 rocprofv3 --kernel-trace -d $O/prof_cli -o trace -- python3 $R/tools/cli_prefetch_probe.py 4 8 120 > $O/${tag}_cli_prof.log 2>&1
db=$(ls $O/prof_cli/*.db 2>/dev/null | head -1)
echo "# the prefetch run under rocprofv3 --kernel-trace: idle gaps of the GPU (union of all streams) longer than 1 ms, after the model build"
python3 - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kt = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall() if r[0].startswith('kernels')][0]
rows = cur.execute(f"select start, end from {kt} order by start").fetchall()
# the CLI run = everything behind the last gap > 1 s (clip generation on the host) -- or the whole trace
end = rows[0][1]; start_i = 0
for i in range(1, len(rows)):
    if rows[i][0] - end > 1e9: start_i = i
    end = max(end, rows[i][1])
rows = rows[start_i:]
end = rows[0][1]; gaps = []; busy = 0
for s, e in rows[1:]:
    if s > end: gaps.append((s - end) / 1e6)
    if e > end: busy += (e - max(s, end)) / 1e6; end = e
span = (rows[-1][1] - rows[0][0]) / 1e6
big = sorted([g for g in gaps if g >= 1.0], reverse=True)
print(f'launches {len(rows)}, span {span:.1f} ms, GPU busy {busy:.1f} ms, idle {span - busy:.1f} ms = {(span - busy) / span * 100:.1f} %')
print(f'gaps >= 1 ms: {len(big)}: ' + ', '.join(f'{g:.1f}' for g in big[:20]) + ' ms')
print(f'gaps 0.2 - 1 ms: {sum(1 for g in gaps if 0.2 <= g < 1.0)} (total {sum(g for g in gaps if 0.2 <= g < 1.0):.1f} ms)')
PY
rm -rf $O/prof_cli
} > $O/${tag}_cli_prefetch.txt 2>&1
cat $O/${tag}_cli_prefetch.txt
