"""In-process A/B of conv kernel variants on the big D shapes: variants are (label, impl, env dict); rounds are interleaved
and the median per (shape, variant) is printed.  usage: pp_ab.py B [rounds]"""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
S = 120000
VARIANTS = [("old", L.IMPL_MFMA, {}), ("pp32", L.IMPL_MFMA_PP, {"GA_GEMM_PP_MFMA": "32"}), ("pp16", L.IMPL_MFMA_PP, {"GA_GEMM_PP_MFMA": "16"}),
            ("pp16hot", L.IMPL_MFMA_PP, {"GA_GEMM_PP_MFMA": "16", "GA_GEMM_PP_NOADV": "1"})]
extra = os.environ.get("PP_AB_VARIANTS")
if extra:  # "label:impl:K=V,K=V;..."
    VARIANTS = []
    for item in extra.split(";"):
        lab, impl, kv = (item.split(":") + ["", ""])[:3]
        VARIANTS.append((lab, int(impl), dict(x.split("=") for x in kv.split(",") if x)))
KEYS = sorted({k for _, _, e in VARIANTS for k in e})
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
cases = []
for sc, s in enumerate((S, S // 2, S // 4)):
    cases += [(f"D{sc}.b2", ConvGeom("conv", 64, 128, 41, 4, 20, 1), s // 4),
              (f"D{sc}.b3", ConvGeom("conv", 128, 256, 41, 4, 20, 1), s // 16), (f"D{sc}.b4", ConvGeom("conv", 256, 512, 41, 4, 20, 1), s // 64)]
torch.manual_seed(0)
tot = {lab: 0.0 for lab, _, _ in VARIANTS}
for name, g, Ln in cases:
    w = torch.randn(*g.wdims, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    # AB_COLD=1 (default): the launches of a timing loop rotate through copies of their operands, > 1 GB in all, so that no launch
    # finds its input in the 256 MB Infinity Cache -- as in the training step (same-operand loops flatter one-workgroup-per-CU kernels)
    per = max(B * Ln * g.cin, B * Lo * g.cout) * 2
    ncopy = max(1, min(8, int(1.2e9 // per) + 1)) if os.environ.get("AB_COLD", "1") == "1" else 1
    xs = [torch.randn(B, Ln, g.cin, device="cuda").bfloat16() for _ in range(ncopy)]
    gys = [torch.randn(B, Lo, g.cout, device="cuda").bfloat16() for _ in range(ncopy)]
    x, gy = xs[0], gys[0]
    rot = [0]
    def nxt(lst):
        rot[0] = (rot[0] + 1) % ncopy
        return lst[rot[0]]
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    flops = 2.0 * B * Lo * g.cout * g.cin * g.k
    for mode in ("fwd", "dgrad"):
        fn = (lambda: ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)) if mode == "fwd" else (lambda: ops.conv_dgrad(g, gy, wd, Ln))
        fnt = (lambda: ops.conv_forward(g, nxt(xs), wf, bias, post_act=L.ACT_LRELU)) if mode == "fwd" else (lambda: ops.conv_dgrad(g, nxt(gys), wd, Ln))
        times = {lab: [] for lab, _, _ in VARIANTS}
        errs = {}
        ref = None
        for rnd in range(ROUNDS):
            for lab, impl, env in VARIANTS:
                for k in KEYS: os.environ.pop(k, None)
                os.environ.update(env)
                ops.set_conv_impl(impl)
                try:
                    if rnd == 0:
                        y = fn().float(); torch.cuda.synchronize()
                        if ref is None: ref = y
                        errs[lab] = ((y - ref).abs().max() / ref.abs().max()).item()
                    times[lab].append(timeit(fnt))
                except Exception as e:
                    errs[lab] = f"{type(e).__name__}"
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
        row = f"{name:6s} {mode:5s} M{g.cout:4d} K{g.cin*g.k:6d} Lout{Lo:6d} |"
        for lab, _, _ in VARIANTS:
            if times[lab]:
                t = statistics.median(times[lab]); tot[lab] += t
                row += f" {lab}: {t*1e3:7.1f}us {flops/t/1e9:5.0f}TF e{errs[lab]:.0e} |"
            else:
                row += f" {lab}: {errs.get(lab)} |"
        print(row, flush=True)
print("total ms:", {k: round(v, 3) for k, v in tot.items()})
