"""In-process A/B of the weight-gradient kernel: flat-address vs buffer-load staging (GA_WGRAD_BUF), bit-compared."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
S = 120000
BB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cases = []
for sc, s in enumerate((S, S // 2, S // 4)):
    cases += [(f"D{sc}.b1", ConvGeom("conv", 32, 64, 41, 4, 20, 1), s, BB), (f"D{sc}.b2", ConvGeom("conv", 64, 128, 41, 4, 20, 1), s // 4, BB),
              (f"D{sc}.b3", ConvGeom("conv", 128, 256, 41, 4, 20, 1), s // 16, BB), (f"D{sc}.b4", ConvGeom("conv", 256, 512, 41, 4, 20, 1), s // 64, BB)]
cases += [("G.in", ConvGeom("conv", 128, 256, 3, 1, 1, 1), 375, 64), ("G.blk", ConvGeom("conv", 256, 256, 3, 1, 8, 8), 375, 64), ("G.out", ConvGeom("conv", 256, 128, 3, 1, 1, 1), 375, 64)]
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = {"0": 0.0, "1": 0.0}
torch.manual_seed(0)
for name, g, Ln, B in cases:
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    res, times = {}, {"0": [], "1": []}
    for rnd in range(3):
        for v in ("0", "1"):
            os.environ["GA_WGRAD_BUF"] = v
            if rnd == 0:
                out = ops.conv_wgrad(g, gy, x)
                res[v] = [t.float().clone() for t in (out if isinstance(out, (tuple, list)) else [out]) if t is not None]
            times[v].append(timeit(lambda: ops.conv_wgrad(g, gy, x)))
    err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res["1"], res["0"]))
    t0, t1 = statistics.median(times["0"]), statistics.median(times["1"])
    tot["0"] += t0; tot["1"] += t1
    print(f"{name:8s} Cin{g.cin:4d} M{g.cout:4d} B{B} Lout{Lo:6d} | flat {t0*1e3:7.1f}us {flops/t0/1e9:5.0f}TF | buf {t1*1e3:7.1f}us {flops/t1/1e9:5.0f}TF | rel diff {err:.1e}", flush=True)
print("total ms", tot)
