"""Per-shape timing of the MFMA conv kernels (forward and data gradient) at the bench batch size."""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
impls = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [L.IMPL_MFMA, L.IMPL_MFMA_WINDOW]
S = 120000
cases = []
for sc, s in enumerate((S, S // 2, S // 4)):
    cases += [(f"D{sc}.b1", ConvGeom("conv", 32, 64, 41, 4, 20, 1), s), (f"D{sc}.b2", ConvGeom("conv", 64, 128, 41, 4, 20, 1), s // 4),
              (f"D{sc}.b3", ConvGeom("conv", 128, 256, 41, 4, 20, 1), s // 16), (f"D{sc}.b4", ConvGeom("conv", 256, 512, 41, 4, 20, 1), s // 64)]
cases += [("G.in", ConvGeom("conv", 128, 256, 3, 1, 1, 1), 375), ("G.blk", ConvGeom("conv", 256, 256, 3, 1, 8, 8), 375),
          ("dec.c7", ConvGeom("cconv", 128, 512, 7), 375), ("dec.t8", ConvGeom("convt", 512, 256, 16, 8), 375),
          ("dec.r3", ConvGeom("cconv", 256, 128, 3), 3000), ("dec.r1", ConvGeom("cconv", 128, 256, 1), 3000),
          ("dec.t5", ConvGeom("convt", 256, 128, 10, 5), 3000), ("dec.t4", ConvGeom("convt", 128, 64, 8, 4), 15000),
          ("dec.t2", ConvGeom("convt", 64, 32, 4, 2), 60000)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
import os
print(f"B={B} nbuf={os.environ.get('GA_GEMM_NBUF')}")
for name, g, Ln in cases:
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    d0, d1, k = g.wdims
    w = torch.randn(d0, d1, k, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k if g.kind != "convt" else 2.0 * B * Ln * g.cout * g.cin * g.k
    row = f"{name:8s} Cin{g.cin:4d} M{g.cout:4d} k{g.k:2d} Lout{Lo:6d} GF{flops/1e9:7.1f} |"
    for impl in impls:
        ops.set_conv_impl(impl)
        try:
            tf = timeit(lambda: ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU))
            td = timeit(lambda: ops.conv_dgrad(g, gy, wd, Ln))
            row += f" impl{impl}: fwd {tf*1e3:7.1f}us {flops/tf/1e9:6.0f}TF  dgrad {td*1e3:7.1f}us {flops/td/1e9:6.0f}TF |"
        except Exception as e:
            row += f" impl{impl}: {type(e).__name__} |"
        finally:
            ops.set_conv_impl(L.IMPL_AUTO)
    print(row, flush=True)
