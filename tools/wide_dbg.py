import torch, torch.nn.functional as F
from gan_amd import ops, _lib as L
from gan_amd.ops import ConvGeom
DEV="cuda"
torch.manual_seed(1)
c=128
gs = [ConvGeom("cconv", c, c, 1), ConvGeom("cconv", c, c // 2, 3), ConvGeom("cconv", c // 2, c, 1)]
packs = [ops.pack_frozen(g, torch.randn(g.cout, g.cin, g.k, device=DEV) * (1.5 / (g.cin * g.k) ** 0.5), None, torch.bfloat16) for g in gs]
bias = [torch.randn(g.cout, device=DEV) * 0.1 for g in gs]
for B, lin in ((2, 1501), (3, 15000), (9, 15000), (9, 64 * 300)):
    x = torch.randn(B, lin, c, device=DEV).bfloat16()
    res = {}
    for fuse in (True, False):
        ops.FUSE_RESUNIT = fuse
        xx = x.clone().requires_grad_(True)
        y = ops.ResBlockFrozen.apply(xx, *packs[0], bias[0], gs[0], *packs[1], bias[1], gs[1], *packs[2], bias[2], gs[2], False, False)
        res[fuse] = (y.detach().float(), y.grad_fn.saved_tensors[1].detach().float())
    href = F.elu(res[False][1]).bfloat16().float()
    bad = ((res[True][1] - href).abs() > 2.0 ** -7 * href.abs().clamp_min(1e-3))
    ybad = (res[True][0] - res[False][0]).abs() > 0.05 * res[False][0].abs().max()
    print(B, lin, "tiles", B * ((lin + 63) // 64), "bad hidden", int(bad.sum()), "bad y", int(ybad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print(" first", idx[:5].tolist(), " last", idx[-3:].tolist())
        rows = idx[:, 1].unique()
        print(" rows mod 64:", sorted(set((rows % 64).tolist()))[:20], "n rows", len(rows), "items", idx[:, 0].unique().tolist(), "chan", sorted(set(idx[:, 2].tolist()))[:10])
        i0 = idx[0].tolist()
        print(" got", res[True][1][i0[0], i0[1], i0[2]].item(), "want", href[i0[0], i0[1], i0[2]].item())
