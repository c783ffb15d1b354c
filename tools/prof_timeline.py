"""Timeline of one step from a rocprofv3 kernel-trace database: start (us from the step's first kernel), duration, queue, name.
usage: prof_timeline.py <db> <out.txt>   (the step = the span between the last two adamw_kernel pairs of the trace)"""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [d[0] for d in c.execute(f"select * from {ks} limit 1").description]
namecol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select d.start, d.end, d.queue_id, d.stream_id, s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[4]]
# a step has two adamw launches (D then G); take the span between the G-adamw of step n-3 and of step n-2 from the end
g_ad = ad[1::2]
mid = len(g_ad) // 2
i0, i1 = g_ad[mid - 1] + 1, g_ad[mid] + 1
t0 = rows[i0][0]
with open(out, "w") as f:
    f.write(f"step wall {(rows[i1 - 1][1] - t0) / 1e3:.1f} us, {i1 - i0} kernels\n")
    for s, e, q, st, n in rows[i0:i1]:
        n = n.replace("void ", "").split("(")[0][:70]
        f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q} s{st}  {n}\n")
print(open(out).readline())
