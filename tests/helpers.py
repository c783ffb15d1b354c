"""Shared test helpers (same procedural inputs as tests/golden/make_golden.py)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle.procfill import proc_input


def make_inputs(B, T, tag, z_style_dim=128):
    q_src = proc_input((B, 128, T), f"{tag}.q_src", 0, 0.9)
    zc = F.normalize(proc_input((B, 128), f"{tag}.zc", 0), dim=-1)
    zs = F.normalize(proc_input((B, z_style_dim), f"{tag}.zs", 0), dim=-1)
    noise = proc_input((B, 32), f"{tag}.noise", 0, 1.7)
    q_real_d = proc_input((B, 128, T), f"{tag}.q_real_d", 0, 0.9)
    q_real_g = proc_input((B, 128, T), f"{tag}.q_real_g", 0, 0.9)
    return q_src, zc, zs, noise, q_real_d, q_real_g


def tsummary(t: torch.Tensor) -> np.ndarray:
    f = t.detach().double().reshape(-1).cpu()
    head = f[:4].tolist() + [0.0] * (4 - min(4, f.numel()))
    tail = f[-4:].tolist() + [0.0] * (4 - min(4, f.numel()))
    return np.array([f.norm().item(), f.sum().item(), f.abs().sum().item()] + head[:4] + tail[:4], dtype=np.float64)


def assert_summary_close(got: torch.Tensor, want: np.ndarray, rtol=1e-4, name="", elem_atol=None, sum_atol=0.0):
    """Compare a tensor with a golden fingerprint [norm, sum, abssum, first4, last4].
    Sums of signed values are compared against the abs-sum scale (they cancel)."""
    g = tsummary(got)
    scale = max(want[2], 1e-12)
    n = max(1, got.numel())
    assert abs(g[0] - want[0]) <= rtol * max(want[0], 1e-12) + sum_atol, f"{name}: norm {g[0]} vs {want[0]}"
    assert abs(g[2] - want[2]) <= rtol * scale + sum_atol, f"{name}: abssum {g[2]} vs {want[2]}"
    assert abs(g[1] - want[1]) <= rtol * scale + sum_atol, f"{name}: sum {g[1]} vs {want[1]}"
    elem = scale / n  # mean |x|
    np.testing.assert_allclose(g[3:], want[3:], rtol=rtol * 10, atol=(rtol * 10 * elem if elem_atol is None else elem_atol), err_msg=name)


def assert_grad_close(got, want, rtol=2e-3, atol_frac=2e-4, max_bad_frac=0.02, l2_tol=1e-2, name=""):
    """Gradient comparison that tolerates LeakyReLU / hinge / |.| kink flips: an activation within fp32 noise of zero
    takes the other branch in one implementation and changes the gradient over its receptive field (observed: one flip
    -> ~430 neighbouring input samples, 1.3 % of the tensor).  Elementwise bound on all but `max_bad_frac` of the
    elements, plus a relative L2 bound on the whole tensor."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    tol = rtol * np.abs(want) + atol_frac * np.abs(want).max()
    bad = np.abs(got - want) > tol
    assert bad.mean() <= max_bad_frac, f"{name}: {bad.mean():.4f} of the elements off"
    l2 = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
    assert l2 <= l2_tol, f"{name}: relative L2 error {l2:.3e}"
