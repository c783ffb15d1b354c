"""Grouped weight gradient (ga_wgrad_multi) of generator-sized layers against one launch per layer; env variants as in wgrad_ab.py
(WG_AB_VARIANTS="label:K=V,...;...", read per launch).  Cold operands: every layer has its own tensors (0.5 GB in all)."""
import os, sys, statistics, ctypes as C, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
n, B, T, Cc = 22, 64, 375, 256
torch.manual_seed(0)
geoms = [ConvGeom("conv", Cc, Cc, 3, 1, 2 ** (i % 4), 2 ** (i % 4)) for i in range(n)]
xs = [torch.randn(B, T, Cc, device="cuda").bfloat16() for _ in range(n)]
gs = [torch.randn(B, T, Cc, device="cuda").bfloat16() for _ in range(n)]
flops = sum(2.0 * B * T * g.cout * g.cin * g.k for g in geoms)
def single():
    return [ops.conv_wgrad(g, ga, x, want_bias=True) for g, ga, x in zip(geoms, gs, xs)]
def multi():
    sizes = [g.cout * g.k * g.cin for g in geoms]
    slab = torch.zeros(sum(sizes) + n * Cc, device="cuda")
    items = (L.WgradParams * n)()
    off, outs = 0, []
    for i, (it, g, ga, x) in enumerate(zip(items, geoms, gs, xs)):
        dw = slab[off:off + sizes[i]]; off += sizes[i]
        db = slab[sum(sizes) + i * Cc: sum(sizes) + (i + 1) * Cc]
        ops._fill(it, g=ga.data_ptr(), x=x.data_ptr(), dw=dw.data_ptr(), dbias=db.data_ptr(), B=B, Lin=T, Cin=g.cin, M=g.cout, Lout=T, J=g.k,
                  in_stride=1, tap_step=g.dilation, off0=-g.padding, LgTot=T, g_stride=1, g_phase=0)
        outs.append((dw.view(g.cout, g.k, g.cin), db))
    L.check(L.lib().ga_wgrad_multi(C.cast(items, C.c_void_p), n, L.DT_BF16, L.stream()), "ga_wgrad_multi")
    return outs
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = single()
VARS = [("default", {})]
if os.environ.get("WG_AB_VARIANTS"):
    VARS = [(it.split(":")[0], dict(kv.split("=") for kv in it.split(":")[1].split(",") if kv)) for it in os.environ["WG_AB_VARIANTS"].split(";")]
KEYS = sorted({k for _, e in VARS for k in e})
t1 = statistics.median(timeit(single) for _ in range(3))
print(f"one launch per layer: {t1 * 1e3:.0f} us ({flops / t1 / 1e9:.0f} TFLOP/s)")
for lab, env in VARS:
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    got = multi()
    err = max(float((a[0] - b[0]).abs().max() / b[0].abs().max()) for a, b in zip(got, ref))
    errb = max(float((a[1] - b[1]).abs().max() / b[1].abs().max()) for a, b in zip(got, ref))
    t = statistics.median(timeit(multi) for _ in range(3))
    print(f"grouped [{lab}]: {t * 1e3:.0f} us ({flops / t / 1e9:.0f} TFLOP/s)  max rel diff dw {err:.1e} db {errb:.1e}")
