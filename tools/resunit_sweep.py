"""Random shapes: one-launch residual unit (forward, tail form, backward) against the three-launch form."""
import random, sys, torch
import torch.nn.functional as F
from gan_amd import ops, _lib as L
from gan_amd.ops import ConvGeom
DEV = "cuda"
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
def rel(a, b):
    return float((a.float() - b.float()).abs().max() / max(float(b.float().abs().max()), 1e-12))
worst = [0, 0, 0]
specials = [3, 4, 5, 29, 30, 31, 32, 33, 61, 62, 63, 64, 65, 66, 121, 122, 123, 124, 125, 126, 127, 128, 129, 130, 244, 245, 256, 257]
for it in range(160):
    c = random.choice((32, 64))
    lin = random.choice(specials) if it % 2 == 0 else random.randint(3, 3000)
    B = random.randint(1, 5)
    gs = {"sc": ConvGeom("cconv", c, c, 1), "b1": ConvGeom("cconv", c, c // 2, 3), "b3": ConvGeom("cconv", c // 2, c, 1), "last": ConvGeom("cconv", 32, 1, 7)}
    packs = {n: ops.pack_frozen(g, torch.randn(g.cout, g.cin, g.k, device=DEV) * (1.5 / (g.cin * g.k) ** 0.5), None, torch.bfloat16) for n, g in gs.items()}
    bias = {n: torch.randn(g.cout, device=DEV) * 0.1 for n, g in gs.items()}
    x = torch.randn(B, lin, c, device=DEV).bfloat16()
    gy = torch.randn(B, lin, c, device=DEV).bfloat16()
    outs = {}
    for mode in ("three", "one"):
        ops.FUSE_RESUNIT = ops.FUSE_RESUNIT_BWD = mode == "one"
        xx = x.clone().requires_grad_(True)
        y = ops.ResBlockFrozen.apply(xx, *packs["sc"], bias["sc"], gs["sc"], *packs["b1"], bias["b1"], gs["b1"], *packs["b3"], bias["b3"], gs["b3"], False, False)
        gx, = torch.autograd.grad(y, xx, gy)
        outs[mode] = (y.detach(), gx)
    ey, eg = rel(outs["one"][0], outs["three"][0]), rel(outs["one"][1], outs["three"][1])
    et = 0.0
    if c == 32 and lin >= 13:
        wav1, _ = ops.resunit_forward(x, packs["sc"][0], bias["sc"], packs["b1"][0], bias["b1"], packs["b3"][0], bias["b3"], want_t=False, last=(packs["last"][0], bias["last"]))
        wav2 = ops.conv_forward(gs["last"], outs["one"][0], packs["last"][0], bias["last"], pre_act=L.ACT_ELU, out_f32=True)
        et = rel(wav1, wav2)
    worst = [max(worst[0], ey), max(worst[1], eg), max(worst[2], et)]
    bad = ey > 1e-2 or eg > 1e-2 or et > 1e-5 or not all(torch.isfinite(t).all() for t in (outs["one"][0], outs["one"][1]))
    if bad:
        print("BAD", c, lin, B, ey, eg, et, flush=True)
        sys.exit(1)
print("ok: worst y %.2e gx %.2e tail %.2e" % tuple(worst))
