"""MI355X-native drop-in for the MR-STFT part of the reference's `lab 3/src/lab3_losses.py`: `multi_resolution_stft_loss`
(:16-71), the mel-trajectory sibling of the codec track's `_waveform_mrstft_loss`: every mel band
is a 1-D signal over time, [B, n_mels, T] -> B*n_mels signals of T samples, resolutions (64,16,64), (128,32,128), (256,64,256),
magnitudes floored as |X| + 1e-8 (the waveform loss clamps instead).  Same kernels as the waveform loss: reflect pad + framed
DFT-basis GEMM on `ga_conv` (bf16 hi/lo split in bf16 mode), fused magnitude / log / three-sum reduction (`ga_stft_loss_*_mode`
with GA_STFT_MAG_ADD), scalar tail `ga_stft_combine`.

The other functions of that file (:10-13, :74-286) are the mel track's stage-2 regularisers (section 8f next-4): per-frame statistics
of a [B, n_mels, T] dB mel (a few hundred KB per batch), written here as device tensor expressions with the reference's names,
arguments, defaults and ValueErrors; autograd differentiates them.  They are pinned by tests/golden/melstage2_golden.npz.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops


def multi_resolution_stft_loss(mel_pred: torch.Tensor, mel_true: torch.Tensor,
                               resolutions: Iterable[Tuple[int, int, int]] = ((64, 16, 64), (128, 32, 128), (256, 64, 256)),
                               precision: str = "fp32", sum_hook=None) -> torch.Tensor:
    """reference lab3_losses.py:16-71 (same name, arguments and ValueError).  The gradient flows to `mel_pred`; `mel_true` is
    treated as data (detached), which is how the mel track uses it.  `sum_hook` (data parallel, gan_amd.dp.DataParallel.stft_sum_hook)
    may all-reduce a resolution's three sums so that the spectral-convergence ratio is the global batch's."""
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
        # (the split-bf16 GEMM stages 16-byte pieces of the 3 x hop framed channels and its data gradient has hop output channels:
        # hops below 16 or not a multiple of 8 -- toy sizes -- take the exact fp32 kernel)
        sums = ops.StftSums.apply(x, y, n_fft, hop, win, precision == "bf16" and hop >= 16 and hop % 8 == 0, True)
        count = float(b * n_mels * (n_fft // 2 + 1) * (1 + t // hop))
        if sum_hook is not None:
            sums, count = sum_hook(sums, count)
        per_res.append(sums)
        inv_counts.append(1.0 / count)
    if not per_res:
        return ops.reduce_sum(mel_pred, mel_true.detach(), L.RED_ABSDIFF, 1.0 / mel_pred.numel())
    return ops.StftCombine.apply(torch.stack(per_res), ops.const_vec(inv_counts, x.device))


def cosine_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """reference lab3_losses.py:10-13: 1 - cos of the L2-normalised rows."""
    return 1.0 - (F.normalize(a, dim=-1) * F.normalize(b, dim=-1)).sum(dim=-1)


def _power(mel_db: torch.Tensor) -> torch.Tensor:
    return torch.pow(10.0, mel_db / 10.0).clamp_min(1e-10)


def _band_share(p: torch.Tensor, rows: slice) -> torch.Tensor:
    """[B]: time-mean energy of the mel rows `rows` over the time-mean total energy (floored at 1e-10)."""
    return p[:, rows, :].sum(dim=1).mean(dim=1) / p.sum(dim=1).mean(dim=1).clamp_min(1e-10)


def spectral_flatness_penalty_from_db(mel_db: torch.Tensor, target_max: float = 0.12) -> torch.Tensor:
    """reference :74-85: per-frame geometric / arithmetic mean over the mel bins, hinged at target_max."""
    p = _power(mel_db)
    flat = torch.exp(torch.log(p).mean(dim=1)) / p.mean(dim=1).clamp_min(1e-10)
    return F.relu(flat - float(target_max)).mean()


def timbre_balance_penalty_from_db(mel_db: torch.Tensor, hf_top_frac: float = 0.25, lf_bottom_frac: float = 0.25,
                                   hf_max_ratio: float = 0.45, lf_min_ratio: float = 0.10) -> torch.Tensor:
    """reference :88-110: too much energy in the top bins, or too little in the bottom ones."""
    p = _power(mel_db)
    n = int(p.shape[1])
    top, bot = max(1, int(round(n * float(hf_top_frac)))), max(1, int(round(n * float(lf_bottom_frac))))
    return (F.relu(_band_share(p, slice(n - top, n)) - float(hf_max_ratio)) + F.relu(float(lf_min_ratio) - _band_share(p, slice(0, bot)))).mean()


def _fill_outside(mel_db: torch.Tensor, lo: int, hi: int, fill_db: float) -> torch.Tensor:
    if mel_db.ndim != 3:
        raise ValueError(f"Expected [B, n_mels, T], got {tuple(mel_db.shape)}")
    keep = torch.zeros(mel_db.shape[1], dtype=torch.bool, device=mel_db.device)
    keep[lo:hi] = True
    return torch.where(keep.view(1, -1, 1), mel_db, mel_db.new_full((), float(fill_db)))


def lowpass_mel_db(mel_db: torch.Tensor, keep_bins: int = 80, fill_db: float = -80.0) -> torch.Tensor:
    """reference :113-124: bins from keep_bins up are set to fill_db (no gradient there)."""
    n = int(mel_db.shape[1]) if mel_db.ndim == 3 else 0
    return _fill_outside(mel_db, 0, int(max(1, min(int(keep_bins), n))), fill_db)


def estimate_lowpass_keep_bins(n_mels: int, sample_rate: int, cutoff_hz: float, fmin: float = 20.0, fmax: Optional[float] = None) -> int:
    """reference :127-160: how many of n_mels uniformly spaced mel centres (HTK scale, fmin..fmax) lie at or below cutoff_hz."""
    if n_mels <= 1:
        return 1
    hi = float(sample_rate) * 0.5 if fmax is None else float(fmax)
    lo = float(max(1e-3, fmin))
    cutoff = float(max(lo, min(cutoff_hz, hi)))
    m_lo, m_hi = 2595.0 * math.log10(1.0 + lo / 700.0), 2595.0 * math.log10(1.0 + hi / 700.0)
    keep = sum(1 for i in range(int(n_mels))
               if 700.0 * (10.0 ** ((m_lo + (m_hi - m_lo) * (float(i) / float(max(1, n_mels - 1)))) / 2595.0) - 1.0) <= cutoff)
    return int(max(1, min(n_mels, keep)))


def bandpass_mel_db(mel_db: torch.Tensor, low_bin: int = 8, high_bin: int = 56, fill_db: float = -80.0) -> torch.Tensor:
    """reference :163-182: bins outside [low_bin, high_bin) are set to fill_db."""
    n = int(mel_db.shape[1]) if mel_db.ndim == 3 else 1
    lo = int(max(0, min(int(low_bin), n - 1)))
    return _fill_outside(mel_db, lo, int(max(lo + 1, min(int(high_bin), n))), fill_db)


def _bin_weighted_l1(mel_pred: torch.Tensor, mel_true: torch.Tensor, split: int, below: float, above: float) -> torch.Tensor:
    if mel_pred.shape != mel_true.shape:
        raise ValueError(f"Shape mismatch: {tuple(mel_pred.shape)} vs {tuple(mel_true.shape)}")
    w = torch.full((int(mel_pred.shape[1]),), float(above), device=mel_pred.device, dtype=mel_pred.dtype)
    w[:split] = float(below)
    return torch.mean(torch.abs(mel_pred - mel_true) * w.view(1, -1, 1))


def frequency_weighted_l1_loss(mel_pred: torch.Tensor, mel_true: torch.Tensor, split_bin: int = 80, low_gain: float = 5.0,
                               high_gain: float = 0.5) -> torch.Tensor:
    """reference :185-203: L1 with low_gain on the bins below split_bin and high_gain on the rest."""
    n = int(mel_pred.shape[1])
    return _bin_weighted_l1(mel_pred, mel_true, int(max(1, min(int(split_bin), n))), low_gain, high_gain)


def spectral_tilt_penalty_from_db(mel_db: torch.Tensor, split_frac: float = 0.5, hf_to_lf_max_ratio: float = 0.7) -> torch.Tensor:
    """reference :206-220: mean power of the upper bins over that of the lower ones, hinged."""
    p = _power(mel_db)
    n = int(p.shape[1])
    split = int(max(1, min(n - 1, round(n * float(split_frac)))))
    ratio = p[:, split:, :].mean(dim=(1, 2)) / (p[:, :split, :].mean(dim=(1, 2)) + 1e-10)
    return F.relu(ratio - float(hf_to_lf_max_ratio)).mean()


def zcr_proxy_penalty_from_db(mel_db: torch.Tensor, target_max: float = 0.18, smooth_k: float = 8.0) -> torch.Tensor:
    """reference :223-240: soft sign flips of the log-energy envelope's slope, hinged."""
    env = torch.log(_power(mel_db).mean(dim=1) + 1e-10)
    if env.shape[1] < 3:
        return torch.zeros((), device=mel_db.device, dtype=mel_db.dtype)
    s = torch.tanh(float(smooth_k) * (env[:, 1:] - env[:, :-1]))
    return F.relu(F.relu(-(s[:, 1:] * s[:, :-1])).mean(dim=1) - float(target_max)).mean()


def highpass_recon_l1_loss(mel_pred: torch.Tensor, mel_true: torch.Tensor, low_cut_bin: int = 12, gain: float = 2.0) -> torch.Tensor:
    """reference :243-260: L1 on the bins from low_cut_bin up (weight `gain`; zero below)."""
    n = int(mel_pred.shape[1])
    return _bin_weighted_l1(mel_pred, mel_true, int(max(0, min(int(low_cut_bin), n - 1))), 0.0, gain)


def hf_excess_penalty_from_db(mel_fake_db: torch.Tensor, mel_real_db: torch.Tensor, top_frac: float = 0.20,
                              max_ratio_multiplier: float = 1.20) -> torch.Tensor:
    """reference :263-285: the fake's top-band energy share above max_ratio_multiplier times the real's."""
    if mel_fake_db.shape != mel_real_db.shape:
        raise ValueError(f"Shape mismatch: {tuple(mel_fake_db.shape)} vs {tuple(mel_real_db.shape)}")
    n = int(mel_fake_db.shape[1])
    top = int(max(1, min(n, round(n * float(top_frac)))))
    rows = slice(n - top, n)
    return F.relu(_band_share(_power(mel_fake_db), rows) - _band_share(_power(mel_real_db), rows) * float(max_ratio_multiplier)).mean()


__all__ = ["multi_resolution_stft_loss", "cosine_distance", "spectral_flatness_penalty_from_db", "timbre_balance_penalty_from_db",
           "lowpass_mel_db", "estimate_lowpass_keep_bins", "bandpass_mel_db", "frequency_weighted_l1_loss", "spectral_tilt_penalty_from_db",
           "zcr_proxy_penalty_from_db", "highpass_recon_l1_loss", "hf_excess_penalty_from_db"]
