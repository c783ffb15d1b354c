import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_melgraph as T
for period, sizes in ((1, [4]*10), (1, [4,4,4,4,4,2,4,4,4,4])):
    G0, D0, cfg = T._setup(period)
    batches = T._batches(10, sizes)
    a = T._run(G0, D0, cfg, batches, False)[4]
    b = T._run(G0, D0, cfg, batches, False)[4]
    c = T._run(G0, D0, cfg, batches, True)[4]
    rel = lambda x, y: max(abs(p - q) / max(abs(q), 1e-9) for p, q in zip(x, y))
    print("sizes", sizes)
    print("  eager vs eager :", " ".join(f"{rel(x, y):.1e}" for x, y in zip(b, a)))
    print("  graph vs eager :", " ".join(f"{rel(x, y):.1e}" for x, y in zip(c, a)))
