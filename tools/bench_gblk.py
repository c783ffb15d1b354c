"""G-block-sized conv (N = 4500 rows, M = 256): time against K depth, to separate fixed cost from the k-loop."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("NBUF", os.environ.get("GA_GEMM_NBUF"), "DEEP", os.environ.get("GA_GEMM_DEEP_MAX"))
for B, Ln in ((12, 375), (48, 375)):
    for k in (1, 3, 7, 15, 31):
        g = ConvGeom("conv", 256, 256, k, 1, k // 2, 1)
        x = torch.randn(B, Ln, 256, device="cuda").bfloat16()
        w = torch.randn(256, 256, k, device="cuda") / (256 * k) ** 0.5
        bias = torch.randn(256, device="cuda")
        wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
        y = torch.empty(B, Ln, 256, device="cuda", dtype=torch.bfloat16)
        p = ops.fwd_params(g, B, Ln)
        def run():
            ops._launch_conv(p, x, wf, y, bias=bias)
        t = timeit(run)
        print(f"B{B} k{k:2d} ksteps {k*4:3d}: {t:6.1f} us  {2.0*B*Ln*256*256*k/t/1e6:6.0f} TF", flush=True)
