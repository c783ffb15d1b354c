"""MI355X-native drop-in for the MR-STFT part of the reference's `lab 3/src/lab3_losses.py`: `multi_resolution_stft_loss`
(:16-71), the mel-trajectory sibling of the codec track's `_waveform_mrstft_loss`: every mel band
is a 1-D signal over time, [B, n_mels, T] -> B*n_mels signals of T samples, resolutions (64,16,64), (128,32,128), (256,64,256),
magnitudes floored as |X| + 1e-8 (the waveform loss clamps instead).  Same kernels as the waveform loss: reflect pad + framed
DFT-basis GEMM on `ga_conv` (bf16 hi/lo split in bf16 mode), fused magnitude / log / three-sum reduction (`ga_stft_loss_*_mode`
with GA_STFT_MAG_ADD), scalar tail `ga_stft_combine`.  The other penalties of that file belong to the mel-track GAN (section 8f
next-4) and are not here.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import torch

from . import _lib as L
from . import ops


def multi_resolution_stft_loss(mel_pred: torch.Tensor, mel_true: torch.Tensor,
                               resolutions: Iterable[Tuple[int, int, int]] = ((64, 16, 64), (128, 32, 128), (256, 64, 256)),
                               precision: str = "fp32") -> torch.Tensor:
    """reference lab3_losses.py:16-71 (same name, arguments and ValueError).  The gradient flows to `mel_pred`; `mel_true` is
    treated as data (detached), which is how the mel track uses it."""
    if mel_pred.shape != mel_true.shape:
        raise ValueError(f"Shape mismatch: {mel_pred.shape} vs {mel_true.shape}")
    L.require_cuda(mel_pred, mel_true)
    b, n_mels, t = mel_pred.shape
    x = mel_pred.reshape(b * n_mels, t)
    y = mel_true.reshape(b * n_mels, t).detach()
    per_res, inv_counts = [], []
    for n_fft, hop, win in resolutions:
        n_fft, hop, win = int(n_fft), int(hop), int(win)
        if t < n_fft:
            continue
        sums = ops.StftSums.apply(x, y, n_fft, hop, win, precision == "bf16", True)
        per_res.append(sums)
        inv_counts.append(1.0 / float(b * n_mels * (n_fft // 2 + 1) * (1 + t // hop)))
    if not per_res:
        return ops.reduce_sum(mel_pred, mel_true.detach(), L.RED_ABSDIFF, 1.0 / mel_pred.numel())
    return ops.StftCombine.apply(torch.stack(per_res), ops.const_vec(inv_counts, x.device))


__all__ = ["multi_resolution_stft_loss"]
