#!/usr/bin/env python3
"""Golden vectors for `multi_resolution_stft_loss` (lab 3/src/lab3_losses.py:16-71), FROM THE REFERENCE.

Run in the build container only:   python tests/golden/make_golden_mel.py      # writes tests/golden/mel_golden.npz

Imports the reference's own function (the module is torch-only) and records, on procedural mel trajectories: the loss at the
mel track's shape [B, 96, 256] (all three resolutions), a short one (T = 100: the 128- and 256-point resolutions are skipped),
the L1 fallback (T = 40), a near-silent input (exercises |X| + 1e-8), and the gradient w.r.t. the prediction.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/lab 3")

from src import lab3_losses as RL  # noqa: E402

from oracle.procfill import proc_input  # noqa: E402


def main():
    store = {}
    x = proc_input((2, 96, 256), "mel.pred", 0, 0.8).requires_grad_(True)
    y = proc_input((2, 96, 256), "mel.true", 0, 0.8)
    l = RL.multi_resolution_stft_loss(x, y)
    store["full"] = np.float64(l.item())
    g, = torch.autograd.grad(l, x)
    store["full_grad"] = g.numpy()[:, ::7, ::5]
    store["full_grad_norm"] = np.float64(g.double().norm().item())
    store["short100"] = np.float64(RL.multi_resolution_stft_loss(x[:, :, :100], y[:, :, :100]).item())
    store["fallback40"] = np.float64(RL.multi_resolution_stft_loss(x[:, :, :40], y[:, :, :40]).item())
    z = torch.zeros(1, 4, 256)
    z[0, 1, 30] = 1e-4
    store["silence"] = np.float64(RL.multi_resolution_stft_loss(z, y[:1, :4]).item())
    path = HERE / "mel_golden.npz"
    np.savez_compressed(path, **store)
    print("wrote", path, {k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in store.items()})


if __name__ == "__main__":
    main()
