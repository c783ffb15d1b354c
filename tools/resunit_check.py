"""Timing of ga_resunit_fwd against the three launches it replaces (bench shapes)."""
import sys, torch
from gan_amd import ops, _lib as L
from gan_amd.ops import ConvGeom
DEV = "cuda"
def run(c, lin, B, want_grad):
    g_sc, g1, g3 = ConvGeom("cconv", c, c, 1), ConvGeom("cconv", c, c // 2, 3), ConvGeom("cconv", c // 2, c, 1)
    packs = [ops.pack_frozen(g, (torch.randn(g.cout, g.cin, g.k, device=DEV) * (1.5 / (g.cin * g.k) ** 0.5)), None, torch.bfloat16) for g in (g_sc, g1, g3)]
    bias = [torch.randn(g.cout, device=DEV) * 0.1 for g in (g_sc, g1, g3)]
    x = torch.randn(B, lin, c, device=DEV).bfloat16()
    res = {}
    for fuse in (False, True):
        ops.FUSE_RESUNIT = fuse
        def f():
            if want_grad:
                xx = x.detach().requires_grad_(True)
                return ops.ResBlockFrozen.apply(xx, *packs[0], bias[0], g_sc, *packs[1], bias[1], g1, *packs[2], bias[2], g3, False, False)
            with torch.no_grad():
                return ops.ResBlockFrozen.apply(x, *packs[0], bias[0], g_sc, *packs[1], bias[1], g1, *packs[2], bias[2], g3, False, False)
        for _ in range(3): y = f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): y = f()
        e1.record(); torch.cuda.synchronize()
        res[fuse] = e0.elapsed_time(e1) / 10
    gb = 2 * B * lin * c * 2 / 1e9 + (B * lin * c / 1e9 if want_grad else 0)
    print(f"C={c} L={lin} B={B} grad={want_grad}: three launches {res[False]*1e3:.0f} us, one launch {res[True]*1e3:.0f} us ({gb/res[True]:.2f} TB/s of x+y{'+t' if want_grad else ''})", flush=True)
def run_bwd(c, lin, B):
    g_sc, g1, g3 = ConvGeom("cconv", c, c, 1), ConvGeom("cconv", c, c // 2, 3), ConvGeom("cconv", c // 2, c, 1)
    packs = [ops.pack_frozen(g, (torch.randn(g.cout, g.cin, g.k, device=DEV) * (1.5 / (g.cin * g.k) ** 0.5)), None, torch.bfloat16) for g in (g_sc, g1, g3)]
    bias = [torch.randn(g.cout, device=DEV) * 0.1 for g in (g_sc, g1, g3)]
    x = torch.randn(B, lin, c, device=DEV).bfloat16()
    gy = torch.randn(B, lin, c, device=DEV).bfloat16()
    ops.FUSE_RESUNIT = True
    xx = x.detach().requires_grad_(True)
    y = ops.ResBlockFrozen.apply(xx, *packs[0], bias[0], g_sc, *packs[1], bias[1], g1, *packs[2], bias[2], g3, False, False)
    res = {}
    for fuse in (False, True):
        ops.FUSE_RESUNIT_BWD = fuse
        for _ in range(3): g, = torch.autograd.grad(y, xx, gy, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g, = torch.autograd.grad(y, xx, gy, retain_graph=True)
        e1.record(); torch.cuda.synchronize()
        res[fuse] = e0.elapsed_time(e1) / 10
    gb = 3.5 * B * lin * c * 2 / 1e9
    print(f"bwd C={c} L={lin} B={B}: three launches + fix-up {res[False]*1e3:.0f} us, one launch {res[True]*1e3:.0f} us ({gb/res[True]:.2f} TB/s of gy+t+x+gx)", flush=True)
if "bwd" in sys.argv:
    for c, lin in ((32, 120000), (64, 60000)):
        for B in (64, 12):
            run_bwd(c, lin, B)
    sys.exit(0)
for c, lin in ((128, 15000), (32, 120000), (64, 60000)):
    for B, wg in ((192, False), (64, True), (36, False), (12, True)):
        run(c, lin, B, wg)
