import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:44], r["Grid_Size"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, g), (n, v) in agg.items():
    if n >= 3 and ("conv_" in k or "wgrad" in k):
        print(f"{k:44s} grid {g:>9s} x{n}: {sys.argv[2]} {2048.0 * v / n / 1e6 if sys.argv[2] == 'read' else 1024.0 * v / n / 1e6:9.1f} MB/launch")
