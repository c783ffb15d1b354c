"""Decoder-tail (HBM-bound) conv shapes and a pure streaming kernel for calibration."""
import sys, os, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = int(sys.argv[1]) if len(sys.argv) > 1 else 36
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(f"B={B} GA_GEMM_NBUF={os.environ.get('GA_GEMM_NBUF')}")
x = torch.randn(B, 60000, 64, device="cuda").bfloat16()
t = timeit(lambda: ops.act_fwd(x, L.ACT_ELU))
print(f"act_fwd elu {x.numel()*2/1e6:.0f} MB in+out each: {t*1e3:.1f} us  {2*x.numel()*2/t/1e9:.2f} TB/s")
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print(f"torch copy: {t*1e3:.1f} us  {2*x.numel()*2/t/1e9:.2f} TB/s")
cases = [("t2", ConvGeom("convt", 64, 32, 4, 2), 60000), ("r3_32", ConvGeom("cconv", 32, 16, 3), 120000), ("r1_32", ConvGeom("cconv", 16, 32, 1), 120000),
         ("sc_32", ConvGeom("cconv", 32, 32, 1), 120000), ("r3_64", ConvGeom("cconv", 64, 32, 3), 60000), ("r1_64", ConvGeom("cconv", 32, 64, 1), 60000),
         ("sc_64", ConvGeom("cconv", 64, 64, 1), 60000), ("t4", ConvGeom("convt", 128, 64, 8, 4), 15000), ("last", ConvGeom("cconv", 32, 1, 7), 120000),
         ("d0", ConvGeom("conv", 1, 32, 15, 1, 7, 1), 120000)]
for name, g, Ln in cases:
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    d0, d1, k = g.wdims
    w = torch.randn(d0, d1, k, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    byts = (x.numel() + gy.numel()) * 2
    tf = timeit(lambda: ops.conv_forward(g, x, wf, bias))
    td = timeit(lambda: ops.conv_dgrad(g, gy, wd, Ln))
    print(f"{name:6s} Cin{g.cin:4d} M{g.cout:4d} k{g.k:2d} Lout{Lo:7d} {byts/1e6:6.0f} MB | fwd {tf*1e3:7.1f}us {byts/tf/1e9:5.2f} TB/s | dgrad {td*1e3:7.1f}us {byts/td/1e9:5.2f} TB/s", flush=True)
