"""GPU: gan_amd.lab3_losses.multi_resolution_stft_loss (the mel-trajectory sibling of the waveform MR-STFT loss, reference
lab 3/src/lab3_losses.py:16-71) against the vectors the reference produced (tests/golden/mel_golden.npz): fp32 mode rtol 1e-4 on
the value, the documented looser bound on the ill-conditioned log-magnitude gradient; bf16 mode (hi/lo split GEMM) within 2e-4."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd.lab3_losses import multi_resolution_stft_loss
from oracle.procfill import proc_input
from tests.helpers import assert_grad_close

GOLD = Path(__file__).resolve().parent / "golden" / "mel_golden.npz"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_mel_mrstft_matches_reference(precision):
    with np.load(GOLD) as z:
        gold = {k: z[k] for k in z.files}
    rel = 1e-4 if precision == "fp32" else 2e-4
    x = proc_input((2, 96, 256), "mel.pred", 0, 0.8).cuda().requires_grad_(True)
    y = proc_input((2, 96, 256), "mel.true", 0, 0.8).cuda()
    l = multi_resolution_stft_loss(x, y, precision=precision)
    assert float(l.detach()) == pytest.approx(float(gold["full"]), rel=rel)
    g, = torch.autograd.grad(l, x)
    # bf16 mode runs the backward DFT GEMM on bf16-rounded spectrum gradients (no hi/lo split there): 5e-2, as for the waveform loss
    gtol = 2e-2 if precision == "fp32" else 5e-2
    assert float(g.double().norm()) == pytest.approx(float(gold["full_grad_norm"]), rel=gtol)
    assert_grad_close(g.cpu().numpy()[:, ::7, ::5], gold["full_grad"], rtol=5e-3 if precision == "fp32" else 5e-2, max_bad_frac=0.05,
                      l2_tol=gtol, name="mel mrstft grad")
    with torch.no_grad():
        assert float(multi_resolution_stft_loss(x[:, :, :100], y[:, :, :100], precision=precision)) == pytest.approx(float(gold["short100"]), rel=rel)
        assert float(multi_resolution_stft_loss(x[:, :, :40], y[:, :, :40], precision=precision)) == pytest.approx(float(gold["fallback40"]), rel=1e-4)
        z0 = torch.zeros(1, 4, 256, device="cuda")
        z0[0, 1, 30] = 1e-4
        assert float(multi_resolution_stft_loss(z0, y[:1, :4], precision=precision)) == pytest.approx(float(gold["silence"]), rel=1e-3)
    with pytest.raises(ValueError):
        multi_resolution_stft_loss(x, y[:, :, :100])
