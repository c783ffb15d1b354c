import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
import ctypes as C
B, Ln = 12, 375
for k in (1, 3):
    g = ConvGeom("conv", 256, 256, k, 1, k // 2, 1)
    x = torch.randn(B, Ln, 256, device="cuda").bfloat16()
    w = torch.randn(256, 256, k, device="cuda") / (256 * k) ** 0.5
    bias = torch.randn(256, device="cuda")
    wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
    y = torch.empty(B, Ln, 256, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(4 * 128, device="cuda", dtype=torch.int64)
    p = ops.fwd_params(g, B, Ln)
    p.x, p.w, p.y, p.bias = x.data_ptr(), wf.data_ptr(), y.data_ptr(), bias.data_ptr()
    p.ws = ws.data_ptr()
    for it in range(3):
        L.check(L.lib().ga_conv(C.byref(p), 1, L.IMPL_AUTO, L.stream()), "conv")
        torch.cuda.synchronize()
    t = ws.cpu().view(-1, 4)[:72].double()
    t0 = t[:, 0].min()
    rel = (t - t0) * 10.0 / 1000.0   # 100 MHz ticks -> us
    print(f"k={k}: block start  min {rel[:,0].min():.2f} max {rel[:,0].max():.2f} us")
    print(f"      prologue    mean {(rel[:,1]-rel[:,0]).mean():.2f} max {(rel[:,1]-rel[:,0]).max():.2f}")
    print(f"      k-loop      mean {(rel[:,2]-rel[:,1]).mean():.2f} max {(rel[:,2]-rel[:,1]).max():.2f}")
    print(f"      epilogue    mean {(rel[:,3]-rel[:,2]).mean():.2f} max {(rel[:,3]-rel[:,2]).max():.2f}")
    print(f"      last end    {rel[:,3].max():.2f}")
