"""Probe (GPU box, from the repo root): forward / backward error of the channels-last norm kernels and of torch's own fp32 modules against
the same modules in float64, per shape and kind."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_gpu_cnorm as T

for kind in ("instance", "batch"):
    for shape in ((4, 32, 24, 64), (6, 64, 12, 32), (6, 16, 48, 128), (2, 512, 6, 16)):
        for act in (False, True):
            (y64, g64, m64), (yh, gh, mh) = T._both(kind, shape, torch.float32, act, ref_dtype=torch.float64)
            (y32, g32, m32), _ = T._both(kind, shape, torch.float32, act, ref_dtype=torch.float32)
            r = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
            print(f"{kind:8s} {str(shape):18s} lrelu={int(act)}  y: hip {r(yh, y64):.2e} torch {r(y32, y64):.2e}   gx: hip {r(gh, g64):.2e} torch {r(g32, g64):.2e}   "
                  f"dgamma: hip {r(mh.weight.grad, m64.weight.grad):.2e} torch {r(m32.weight.grad, m64.weight.grad):.2e}   "
                  f"sum(gx)/sum|gx|: hip {float(gh.double().sum().abs() / gh.double().abs().sum()):.1e} torch {float(g32.double().sum().abs() / g32.double().abs().sum()):.1e}")
