"""GPU idle time inside the timed steps of a rocprofv3 kernel trace: union of kernel intervals vs wall.
usage: prof_gaps.py <db> [skip_fraction]   (the first skip_fraction of the trace -- warm-up, builds -- is ignored)"""
import sqlite3, sys
db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t]
cand = kt[0] if kt else None
print("tables:", [t for t in tabs if "kernel" in t][:8])
cur = c.execute(f"select * from {cand} limit 1")
cols = [d[0] for d in cur.description]
print("cols:", cols)
sc = "start" if "start" in cols else [x for x in cols if "start" in x][0]
ec = "end" if "end" in cols else [x for x in cols if "end" in x][0]
rows = c.execute(f"select {sc}, {ec} from {cand} order by {sc}").fetchall()
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= cut]
busy, cur_e, gaps = 0, rows[0][0], []
for s, e in rows:
    if s > cur_e:
        gaps.append(s - cur_e)
        busy += 0
        cur_s = s
    s2 = max(s, cur_e)
    if e > s2:
        busy += e - s2
        cur_e = e
wall = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  wall {wall/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms  idle {100*(wall-busy)/wall:.1f} %")
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g / 1e3, 1) for g in gaps[:12]])
import collections
h = collections.Counter()
for g in gaps:
    h[min(int(g / 1e3 // 5) * 5, 100)] += g
print("idle by gap size (us bucket -> ms):", {k: round(v / 1e6, 2) for k, v in sorted(h.items())})
