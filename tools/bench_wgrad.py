import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
S = 120000
BB = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cases = []
for sc, s in enumerate((S, S // 2, S // 4)):
    cases += [(f"D{sc}.b1", ConvGeom("conv", 32, 64, 41, 4, 20, 1), s, BB), (f"D{sc}.b2", ConvGeom("conv", 64, 128, 41, 4, 20, 1), s // 4, BB),
              (f"D{sc}.b3", ConvGeom("conv", 128, 256, 41, 4, 20, 1), s // 16, BB), (f"D{sc}.b4", ConvGeom("conv", 256, 512, 41, 4, 20, 1), s // 64, BB)]
cases += [("G.in", ConvGeom("conv", 128, 256, 3, 1, 1, 1), 375, 12), ("G.blk", ConvGeom("conv", 256, 256, 3, 1, 8, 8), 375, 12)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, g, Ln, B in cases:
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    t = timeit(lambda: ops.conv_wgrad(g, gy, x))
    print(f"{name:8s} Cin{g.cin:4d} M{g.cout:4d} k{g.k:2d} B{B} Lout{Lo:6d} GF{flops/1e9:7.1f} | wgrad {t*1e3:7.1f}us {flops/t/1e9:6.0f}TF", flush=True)
