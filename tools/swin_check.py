"""Streaming-window kernel vs the register-streaming kernel on the thin multi-tap decoder layers: bit-equality and timing.
usage: swin_check.py B [quick]"""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
CASES = [("cconv", 64, 32, 3, 1, 60000), ("cconv", 32, 16, 3, 1, 120000), ("convt", 64, 32, 4, 2, 60000)]
for kind, cin, cout, k, s, Ln in CASES:
    g = ConvGeom(kind, cin, cout, k, s)
    for Lx in (Ln, 4097, 70):
        Bx = B if Lx == Ln else 3
        x = torch.randn(Bx, Lx, cin, device="cuda").bfloat16()
        w = torch.randn(*g.wdims, device="cuda") / (cin * k) ** 0.5
        bias = torch.randn(cout, device="cuda")
        Lo = g.out_len(Lx)
        gy = torch.randn(Bx, Lo, cout, device="cuda").bfloat16()
        wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
        for mode in ("fwd", "dgrad"):
            fn = (lambda: ops.conv_forward(g, x, wf, bias, pre_act=L.ACT_ELU)) if mode == "fwd" else \
                 (lambda: ops.conv_dgrad(g, gy, wd, Lx, dact_mode=L.DACT_ELU_FROM_IN, dact_src=x))
            nbytes = (x.numel() + gy.numel()) * 2 + (x.numel() * 2 if mode == "dgrad" else 0)
            row = f"{kind} {cin}->{cout} k{k} B{Bx} L{Lx:6d} {mode:5s} {nbytes/1e6:7.1f}MB |"
            ref = None
            for label, impl in (("stream", L.IMPL_MFMA_STREAM), ("swin", L.IMPL_MFMA_STREAMWIN)):
                ops.set_conv_impl(impl)
                try:
                    y = fn().float(); torch.cuda.synchronize()
                    if ref is None: ref = y
                    err = (y - ref).abs().max().item()
                    t = timeit(fn)
                    row += f" {label}: {t*1e3:7.1f}us {nbytes/t/1e9:5.2f}TB/s maxdiff {err:.1e} |"
                except Exception as e:
                    row += f" {label}: {type(e).__name__} {e} |"
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
            print(row, flush=True)
