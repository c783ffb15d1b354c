import torch
from gan_amd import ops, _lib as L
from gan_amd.ops import ConvGeom
DEV="cuda"
c=32
geoms = {"sc": ConvGeom("cconv", c, c, 1), "b1": ConvGeom("cconv", c, c // 2, 3), "b3": ConvGeom("cconv", c // 2, c, 1), "last": ConvGeom("cconv", c, 1, 7)}
pk = {n: ops.pack_frozen(g, torch.randn(g.cout, g.cin, g.k, device=DEV) * (1.5 / (g.cin * g.k) ** 0.5), None, torch.bfloat16)[0] for n, g in geoms.items()}
bd = {n: torch.randn(g.cout, device=DEV) * 0.1 for n, g in geoms.items()}
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (192, 36):
    x = torch.randn(B, 120000, c, device=DEV).bfloat16()
    args = (x, pk["sc"], bd["sc"], pk["b1"], bd["b1"], pk["b3"], bd["b3"])
    t_tail = tm(lambda: ops.resunit_forward(*args, want_t=False, last=(pk["last"], bd["last"])))
    t_unit = tm(lambda: ops.resunit_forward(*args, want_t=False))
    y, _ = ops.resunit_forward(*args, want_t=False)
    t_last = tm(lambda: ops.conv_forward(geoms["last"], y, pk["last"], bd["last"], pre_act=L.ACT_ELU, out_f32=True))
    print(f"B={B}: tail form {t_tail:.0f} us; unit {t_unit:.0f} + closing conv {t_last:.0f} us", flush=True)
