"""Ping-pong conv kernel: correctness vs the gathered-GEMM kernel (PP off) and per-shape timing.  usage: pp_check.py B [ks]"""
import sys, os, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = 120000
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = []
for sc, s in enumerate((S, S // 2)):
    cases += [(f"D{sc}.b2", ConvGeom("conv", 64, 128, 41, 4, 20, 1), s // 4),
              (f"D{sc}.b3", ConvGeom("conv", 128, 256, 41, 4, 20, 1), s // 16), (f"D{sc}.b4", ConvGeom("conv", 256, 512, 41, 4, 20, 1), s // 64)]
torch.manual_seed(0)
for name, g, Ln in cases:
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    d0, d1, k = g.wdims
    w = torch.randn(d0, d1, k, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    for mode in ("fwd", "dgrad"):
        fn = (lambda: ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)) if mode == "fwd" else (lambda: ops.conv_dgrad(g, gy, wd, Ln))
        row = f"{name:6s} {mode:5s} Cin{g.cin:4d} M{g.cout:4d} Lout{Lo:6d} GF{flops/1e9:7.1f} |"
        ref = None
        for label, impl, ks in (("old", L.IMPL_MFMA, None), ("pp1", L.IMPL_MFMA_PP, 1), ("pp2", L.IMPL_MFMA_PP, 2), ("pp4", L.IMPL_MFMA_PP, 4)):
            ops.set_conv_impl(impl); ops.FORCE_KSPLIT = ks
            try:
                y = fn().float()
                torch.cuda.synchronize()
                if ref is None: ref = y
                err = ((y - ref).abs().max() / ref.abs().max()).item()
                t = timeit(fn)
                row += f" {label}: {t*1e3:7.1f}us {flops/t/1e9:6.0f}TF err {err:.1e} |"
            except Exception as e:
                row += f" {label}: {type(e).__name__} {e} |"
            finally:
                ops.set_conv_impl(L.IMPL_AUTO); ops.FORCE_KSPLIT = None
        print(row, flush=True)
