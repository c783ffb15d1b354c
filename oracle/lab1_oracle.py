"""CPU restatement of the Lab1 content path of the G step.  TEST INFRASTRUCTURE ONLY: nothing under gan_amd/ may import it.

  * `wave_to_logmel`: the reference's `_Lab1WaveAdapter` (lab 3/src/lab3_codec_train.py:179-222).  The reference builds it from
    torchaudio (`functional.resample`, `transforms.MelSpectrogram`, `transforms.AmplitudeToDB`), a third-party dependency that is
    ABSENT from /root/reference and from this image (lab 3/requirements.txt lists `torchaudio` unpinned).  What is restated here
    is torchaudio 2.x's published algorithm, in float64 numpy, written independently of gan_amd/lab1_adapter.py:
      - resample: `_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel` (sinc_interp_hann, lowpass_filter_width 64,
        rolloff 0.9475937167399596; gcd-reduced rates, zero pad (width, width + orig), stride-orig correlation, trim to
        ceil(new * S / orig));
      - MelSpectrogram: reflect-padded centred STFT, periodic Hann window = n_fft, power 2, HTK mel filterbank, norm None;
      - AmplitudeToDB("power", top_db 80): 10 log10(max(x, 1e-10)), floor at (max over the whole [B, 96, T] tensor) - 80.
    **Parity unpinned**: the reference holds no fixture for this function and torchaudio cannot be imported here to make one.
  * `chunk_encoder_forward`: the Lab1 `ChunkEncoder` (lab 3/src/lab3_bridge.py:34-76) in eval mode, from a state dict.  Pinned
    by tests/golden/lab1_golden.npz, which tests/golden/make_golden_lab1.py generates from the reference's own class.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def sinc_kernel(orig_freq: int, new_freq: int, width_param: int = 64, rolloff: float = 0.9475937167399596):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(width_param * orig / base))
    k = np.empty((new, 2 * width + orig), dtype=np.float64)
    for p in range(new):
        for j in range(2 * width + orig):
            t = ((-p / new) + (j - width) / orig) * base
            t = min(max(t, -width_param), width_param)
            win = math.cos(t * math.pi / width_param / 2.0) ** 2
            x = t * math.pi
            k[p, j] = (1.0 if x == 0.0 else math.sin(x) / x) * win * (base / orig)
    return k, width, orig, new


_KCACHE: dict = {}


def resample(wave: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    key = (orig_freq, new_freq)
    if key not in _KCACHE:
        _KCACHE[key] = sinc_kernel(orig_freq, new_freq)
    k, width, orig, new = _KCACHE[key]
    B, S = wave.shape
    x = np.concatenate([np.zeros((B, width)), wave.astype(np.float64), np.zeros((B, width + orig))], axis=1)
    taps = k.shape[1]
    frames = (x.shape[1] - taps) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(x, taps, axis=1)[:, ::orig][:, :frames]   # [B, frames, taps]
    y = np.einsum("bft,pt->bfp", win, k).reshape(B, -1)
    return y[:, : int(math.ceil(new * S / orig))]


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m = np.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f = 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    fb = np.zeros((n_freqs, n_mels))
    for j in range(n_mels):
        lo, ce, hi = f[j], f[j + 1], f[j + 2]
        fb[:, j] = np.maximum(0.0, np.minimum((all_freqs - lo) / (ce - lo), (hi - all_freqs) / (hi - ce)))
    return fb


def wave_to_logmel(wav_b_1_t: np.ndarray, codec_sr: int = 24000, lab1_sr: int = 22050, n_frames: int = 256) -> np.ndarray:
    y = np.asarray(wav_b_1_t, dtype=np.float64)[:, 0, :]
    if codec_sr != lab1_sr:
        y = resample(y, codec_sr, lab1_sr)
    n_fft, hop, n_mels = 1024, 256, 96
    yp = np.pad(y, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2.0 * math.pi * n / n_fft)  # periodic Hann
    frames = np.lib.stride_tricks.sliding_window_view(yp, n_fft, axis=1)[:, ::hop]          # [B, T, n_fft]
    power = np.abs(np.fft.rfft(frames * window, axis=-1)) ** 2                              # [B, T, 513]
    mel = np.maximum(power @ mel_filterbank(n_fft // 2 + 1, 20.0, lab1_sr * 0.5, n_mels, lab1_sr), 1e-8).transpose(0, 2, 1)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    t = db.shape[-1]
    if t > n_frames:
        s = (t - n_frames) // 2
        db = db[:, :, s:s + n_frames]
    elif t < n_frames:
        db = np.concatenate([db, np.full((db.shape[0], db.shape[1], n_frames - t), -80.0)], axis=2)
    return db


def chunk_encoder_forward(sd: Dict[str, torch.Tensor], log_mel: torch.Tensor) -> Dict[str, torch.Tensor]:
    """lab3_bridge.py:34-76 in eval mode: 3 x [Conv2d(k3, s2, p1) -> BatchNorm2d (running statistics) -> ReLU] -> global mean ->
    Linear(128, 256) -> ReLU -> content / style heads, each L2-normalised."""
    h = log_mel.unsqueeze(1)
    for i in (0, 3, 6):
        h = F.conv2d(h, sd[f"backbone.{i}.weight"], sd[f"backbone.{i}.bias"], stride=2, padding=1)
        h = F.batch_norm(h, sd[f"backbone.{i + 1}.running_mean"], sd[f"backbone.{i + 1}.running_var"], sd[f"backbone.{i + 1}.weight"],
                         sd[f"backbone.{i + 1}.bias"], training=False, eps=1e-5)
        h = F.relu(h)
    h = h.mean(dim=(2, 3))
    h = F.relu(F.linear(h, sd["shared.weight"], sd["shared.bias"]))
    return {"shared_feat": h, "z_content": F.normalize(F.linear(h, sd["content_head.weight"], sd["content_head.bias"]), dim=-1),
            "z_style": F.normalize(F.linear(h, sd["style_head.weight"], sd["style_head.bias"]), dim=-1)}


def fill_chunk_encoder(sd: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """Procedural weights for the ChunkEncoder topology (shared with make_golden_lab1.py): conv / linear weights at fan-in scale,
    BatchNorm running_var strictly positive."""
    from .procfill import proc_tensor
    out = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            out[k] = proc_tensor(shape, k, seed, 0.4, 1.0)
        elif k.endswith("running_mean"):
            out[k] = proc_tensor(shape, k, seed, 0.2, 0.0)
        elif v.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[k] = proc_tensor(shape, k, seed, float(np.sqrt(3.0 / fan_in)), 0.0)
        elif k.endswith("weight"):
            out[k] = proc_tensor(shape, k, seed, 0.1, 1.0)
        else:
            out[k] = proc_tensor(shape, k, seed, 0.05, 0.0)
    return out
