#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table (like --stats CSV)."""
import sqlite3, sys, re

def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<90} {'calls':>8} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for n, c, s, a, mn, mx in rows[:top]:
        n = re.sub(r'\(.*', '', n)[:90]
        print(f"{n:<90} {c:>8} {s/1e6:>10.2f} {a/1e3:>9.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*s/total:>6.2f}")
    print(f"TOTAL kernel time {total/1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")

if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
