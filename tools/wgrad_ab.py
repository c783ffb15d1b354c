"""In-process A/B of the weight-gradient kernel, results compared against the first variant.  Default: flat-address vs
buffer-load staging (GA_WGRAD_BUF); WG_AB_VARIANTS="label:K=V,K=V;label2:..." sets other environments (read per launch)."""
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
VARS = [("flat", {"GA_WGRAD_BUF": "0"}), ("buf", {"GA_WGRAD_BUF": "1"})]
if os.environ.get("WG_AB_VARIANTS"):
    VARS = [(it.split(":")[0], dict(kv.split("=") for kv in it.split(":")[1].split(",") if kv)) for it in os.environ["WG_AB_VARIANTS"].split(";")]
KEYS = sorted({k for _, e in VARS for k in e})
tot = {lab: 0.0 for lab, _ in VARS}
torch.manual_seed(0)
for name, g, Ln, B in cases:
    Lo = g.out_len(Ln)
    # COLD (default): every launch of a timing loop reads a different copy of its operands, > 1 GB in rotation, so nothing is left in
    # the 256 MB Infinity Cache from the launch before -- as in the training step, where a layer's operands were written long ago
    per = (B * Ln * g.cin + B * Lo * g.cout) * 2
    ncopy = max(1, min(8, int(1.2e9 // per) + 1)) if os.environ.get("AB_COLD", "1") == "1" else 1
    xs = [torch.randn(B, Ln, g.cin, device="cuda").bfloat16() for _ in range(ncopy)]
    gys = [torch.randn(B, Lo, g.cout, device="cuda").bfloat16() for _ in range(ncopy)]
    x, gy = xs[0], gys[0]
    rot = [0]
    def run():
        rot[0] = (rot[0] + 1) % ncopy
        return ops.conv_wgrad(g, gys[rot[0]], xs[rot[0]])
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    res, times = {}, {lab: [] for lab, _ in VARS}
    for rnd in range(3):
        for lab, env in VARS:
            for k in KEYS: os.environ.pop(k, None)
            os.environ.update(env)
            if rnd == 0:
                out = ops.conv_wgrad(g, gy, x)
                res[lab] = [t.float().clone() for t in (out if isinstance(out, (tuple, list)) else [out]) if t is not None]
            times[lab].append(timeit(run))
    row = f"{name:8s} Cin{g.cin:4d} M{g.cout:4d} B{B} Lout{Lo:6d} |"
    for lab, _ in VARS:
        t = statistics.median(times[lab]); tot[lab] += t
        err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res[lab], res[VARS[0][0]]))
        row += f" {lab} {t*1e3:7.1f}us {flops/t/1e9:5.0f}TF d{err:.0e} |"
    print(row, flush=True)
print("total ms", tot)
