"""LDS-window conv kernel vs the gathered-GEMM kernel on the narrow D layers: correctness and timing.  usage: win_check.py B"""
import sys, torch
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
torch.manual_seed(0)
blk = sys.argv[2] if len(sys.argv) > 2 else "b1"
g = ConvGeom("conv", 32, 64, 41, 4, 20, 1) if blk == "b1" else ConvGeom("conv", 64, 128, 41, 4, 20, 1)
for Ln in ((S, S // 2, S // 4, 3001) if blk == "b1" else (S // 4, S // 8, S // 16)):
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    w = torch.randn(*g.wdims, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    h_out = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    for mode in ("fwd", "dgrad"):
        fn = (lambda: ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)) if mode == "fwd" else \
             (lambda: ops.conv_dgrad(g, gy, wd, Ln, dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=h_out))
        row = f"L{Ln:6d} {mode:5s} GF{flops/1e9:7.1f} |"
        ref = None
        for label, impl in (("gather", L.IMPL_MFMA), ("window", L.IMPL_MFMA_LDSWIN)):
            ops.set_conv_impl(impl)
            try:
                y = fn().float(); torch.cuda.synchronize()
                if ref is None: ref = y
                err = ((y - ref).abs().max() / ref.abs().max()).item()
                t = timeit(fn)
                row += f" {label}: {t*1e3:7.1f}us {flops/t/1e9:6.0f}TF err {err:.1e} |"
            except Exception as e:
                row += f" {label}: {type(e).__name__} {e} |"
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)
        print(row, flush=True)
