import sqlite3, csv, sys
db, out, nsteps = sys.argv[1], sys.argv[2], float(sys.argv[3])
c = sqlite3.connect(db)
cur = c.execute("select * from top_kernels")
cols = [d[0] for d in cur.description]
rows = cur.fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f); w.writerow(cols); w.writerows(rows)
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3/nsteps:.2f} ms/step")
for r in rows[:32]:
    print(f"{r[2]/1e3/nsteps:7.2f} ms/step {r[1]/nsteps:7.1f} calls/step {r[3]:8.1f} us  {r[0][:90]}")
