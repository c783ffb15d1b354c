"""How often a generator conv is caught by the side stream's persistent LSTM launch: from a rocprofv3 kernel-trace database, per
lstm_fwd_persist_kernel<3> launch, the longest conv_gemm_kernel<1,2,2,*> launch on the OTHER queue that starts inside it.
usage: prof_stalls.py <db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [d[0] for d in c.execute(f"select * from {ks} limit 1").description]
namecol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select d.start, d.end, d.queue_id, s.{namecol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
lstm = [r for r in rows if "lstm_fwd_persist_kernelILi3" in r[3] or "lstm_fwd_persist_kernel<3>" in r[3]]
res = []
for s, e, q, _ in lstm:
    inside = [(r[1] - r[0]) / 1e3 for r in rows if r[2] != q and s <= r[0] < e and "conv_gemm_kernel" in r[3]]
    first = min([(r[0] - s) / 1e3 for r in rows if r[2] != q and s <= r[0] < e and "conv_gemm_kernel" in r[3]], default=-1)
    # a conv on the other queue that overlaps the LSTM launch for more than 1.5 ms was caught by it, wherever it started
    caught = [((r[0] - s) / 1e3, (r[1] - r[0]) / 1e3) for r in rows if r[2] != q and "conv_gemm_kernel" in r[3] and r[0] < e and r[1] > s
              and (min(r[1], e) - max(r[0], s)) > 1.5e6]
    res.append(((e - s) / 1e3, len(inside), max(inside, default=0.0), first, caught))
for d, n, mx, first, caught in res:
    c = "; CAUGHT: conv started %+.1f us from the LSTM's start, ran %.0f us" % caught[0] if caught else ""
    print(f"LSTM {d:7.1f} us: {n:3d} convs started inside it, longest {mx:7.1f} us, first one {first:6.1f} us after the LSTM's start{c}")
bad = sum(1 for r in res if r[4])
print(f"{bad} of {len(res)} LSTM launches held a conv of the other queue for more than 1.5 ms")
