"""Few-output thin kernels: decoder last conv (32 -> 1, k7, ELU, fp32 out) and D blocks.0 data gradient (32 -> 1, k15).  usage: bench_thin.py"""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
g = ConvGeom("cconv", 32, 1, 7)
w = torch.randn(*g.wdims, device="cuda") / (32 * 7) ** 0.5
wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
bias = torch.randn(1, device="cuda")
for B in (192, 64):
    x = torch.randn(B, 120000, 32, device="cuda").bfloat16()
    t = timeit(lambda: ops.conv_forward(g, x, wf, bias, pre_act=L.ACT_ELU, out_f32=True))
    print(f"last conv fwd B{B}: {t*1e3:7.1f} us  {x.numel()*2/t/1e9:5.2f} TB/s")
    ref = torch.nn.functional.conv1d(torch.nn.functional.pad(torch.nn.functional.elu(x[:2].float()).permute(0, 2, 1), (6, 0), mode="reflect"), w.bfloat16().float(), bias)
    y = ops.conv_forward(g, x[:2].contiguous(), wf, bias, pre_act=L.ACT_ELU, out_f32=True)
    print("   max err vs torch", (y.permute(0, 2, 1) - ref).abs().max().item(), "ref max", ref.abs().max().item())
    gy = torch.randn(B, 120000, 1, device="cuda").bfloat16() if B == 64 else None
    if gy is not None:
        t = timeit(lambda: ops.conv_dgrad(g, gy, wd, 120000, dact_mode=L.DACT_ELU_FROM_IN, dact_src=x))
        print(f"last conv dgrad B{B}: {t*1e3:7.1f} us  {x.numel()*4/t/1e9:5.2f} TB/s")
g0 = ConvGeom("conv", 1, 32, 15, 1, 7, 1)
w0 = torch.randn(*g0.wdims, device="cuda") / 15 ** 0.5
wd0 = ops.pack_weight(g0, w0, g0.dgrad_pack, torch.bfloat16)
for Ln in (120000, 60000, 30000):
    gy = torch.randn(64, Ln, 32, device="cuda").bfloat16()
    t = timeit(lambda: ops.conv_dgrad(g0, gy, wd0, Ln))
    print(f"D blocks.0 dgrad L{Ln}: {t*1e3:7.1f} us  {gy.numel()*2/t/1e9:5.2f} TB/s")
    dx = ops.conv_dgrad(g0, gy[:2].contiguous(), wd0, Ln).float()
    ref = torch.nn.grad.conv1d_input((2, 1, Ln), w0.bfloat16().float(), gy[:2].float().permute(0, 2, 1), padding=7)
    print("   max err vs torch", (dx.permute(0, 2, 1) - ref).abs().max().item(), "ref max", ref.abs().max().item())
