"""From a rocprofv3 kernel-trace database: for the LAST lstm_fwd_persist launch, the kernels of other queues that started or ended
while it was resident (count, total busy time, median duration) -- i.e. what actually co-ran with the persistent LSTM."""
import sqlite3, sys, statistics
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [d[0] for d in c.execute(f"select * from {ks} limit 1").description]
namecol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select d.start, d.end, d.queue_id, s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
lstm = [r for r in rows if "lstm_fwd_persist" in r[3]][-1]
s0, e0, q0, _ = lstm
print(f"LSTM resident {(e0 - s0) / 1e3:.0f} us on queue {q0}")
inside = [r for r in rows if r[2] != q0 and r[0] >= s0 and r[1] <= e0]
straddle = [r for r in rows if r[2] != q0 and r[0] < e0 and r[1] > s0 and not (r[0] >= s0 and r[1] <= e0)]
by = {}
for r in inside:
    by.setdefault(r[3].split("(")[0][:60], []).append((r[1] - r[0]) / 1e3)
for k, v in by.items():
    print(f"  ran entirely inside: {len(v):4d} x {k}  median {statistics.median(v):.1f} us, total {sum(v):.0f} us")
for r in straddle:
    print(f"  straddles: {r[3].split('(')[0][:60]} {(r[1] - r[0]) / 1e3:.0f} us (starts {(r[0] - s0) / 1e3:+.0f} us, ends {(r[1] - e0) / 1e3:+.0f} us relative to the LSTM)")
