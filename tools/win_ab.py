"""In-process A/B of the LDS-window conv kernel (D blocks.1 forward and folded data gradient): GA_CONV_WIN_BUF 0 / 1."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
KEY = sys.argv[2] if len(sys.argv) > 2 else "GA_CONV_WIN_BUF"
S = 120000
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
g = ConvGeom("conv", 32, 64, 41, 4, 20, 1)
for sc, Ln in enumerate((S, S // 2, S // 4)):
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    w = torch.randn(*g.wdims, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    for mode in ("fwd", "dgrad"):
        fn = (lambda: ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)) if mode == "fwd" else (lambda: ops.conv_dgrad(g, gy, wd, Ln))
        ts, ys = {"0": [], "1": []}, {}
        for rnd in range(3):
            for v in ("0", "1"):
                os.environ[KEY] = v
                if rnd == 0: ys[v] = fn().float()
                ts[v].append(timeit(fn))
        t0, t1 = statistics.median(ts["0"]), statistics.median(ts["1"])
        err = float((ys["1"] - ys["0"]).abs().max())
        print(f"D{sc}.b1 {mode:5s} Lout{Lo:6d} | {KEY}=0 {t0*1e3:7.1f}us {flops/t0/1e9:5.0f}TF | =1 {t1*1e3:7.1f}us {flops/t1/1e9:5.0f}TF | max diff {err:.1e}", flush=True)
