"""Waveform -> Lab1 log-mel adapter of the G step's content / Lab1-style terms (reference `_Lab1WaveAdapter`,
lab 3/src/lab3_codec_train.py:179-222), restated without torchaudio.

The reference builds this from three torchaudio pieces, none of which is importable in this image, so what is restated
here is their PUBLISHED algorithm (torchaudio 2.x `functional.resample`, `transforms.MelSpectrogram`,
`transforms.AmplitudeToDB`); no fixture of the reference can pin it here: **parity unpinned** (DESIGN.md section 8b).

  resample 24 kHz -> 22.05 kHz: windowed-sinc polyphase filter, `sinc_interp_hann`, lowpass_filter_width 64, rolloff
      0.9475937167399596 (:200-207): gcd 150 -> 160 input samples per 147 output samples, 308-tap kernels, zero padding
      (width, width + 160), one strided conv, output trimmed to ceil(147 * S / 160).
  MelSpectrogram(sr, n_fft 1024, hop 256, 96 mels, f_min 20, f_max sr/2, power 2, center, reflect pad, periodic Hann,
      HTK mel scale, norm None) (:185-194), clamp_min(1e-8) (:208).
  AmplitudeToDB(stype "power", top_db 80) (:195): 10*log10(max(x, 1e-10)), then clamped from below at (max over the WHOLE
      [B, 96, T] tensor) - 80: torchaudio treats a 3-D input as one multi-channel spectrogram, so the floor is shared by the
      batch.
  crop (centre) or pad (-80) to n_frames (:210-221).

`Lab1WaveAdapter` (the class `train_codec_stage` uses) runs the three contractions on the library's HIP kernels, through the
same C-ABI entry points as the rest of the step, and needs a GPU tensor:

  resample     one `ga_conv`: the zero-padded waveform viewed as rows of 160 samples, two row-taps (320 >= 308 kernel taps),
               147 (padded to 152) output channels = the polyphase branches; exact fp32 FMA kernel
  STFT         `ga_stft_pad` + the DFT-basis `ga_conv` of the MR-STFT loss (bf16 hi/lo split MFMA GEMM, fp32 output)
  mel          `ga_power_spec_fwd` + one k = 1 `ga_conv` (513 -> 96 filterbank), exact fp32

with the data gradients through the same kernels (`ConvFrozen`, `StftSpec`, `PowerSpec` in gan_amd/ops.py).  The dB / top_db /
crop tail is element-wise work on a [B, 96, 431] tensor and stays ordinary torch (autograd gives the floor's gradient path).
`wave_to_logmel_torch` is the same algorithm in plain torch ops (any device): the specification the HIP path is tested against.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn.functional as F


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 64,
                         rolloff: float = 0.9475937167399596, dtype=torch.float32, device="cpu") -> Tuple[torch.Tensor, int, int, int]:
    """[new, 1, 2*width + orig] polyphase kernels of torchaudio's `_get_sinc_resample_kernel` (sinc_interp_hann), computed in
    `dtype` as torchaudio does for a waveform of that dtype.  Returns (kernel, width, orig/gcd, new/gcd)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * float(rolloff)
    width = int(math.ceil(lowpass_filter_width * orig / base_freq))
    idx = torch.arange(-width, width + orig, dtype=dtype, device=device)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=dtype, device=device)[:, None, None] / new + idx
    t = t * base_freq
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t)
    kernels = kernels * window * scale
    return kernels, width, orig, new


def resample(wave: torch.Tensor, kernel: torch.Tensor, width: int, orig: int, new: int) -> torch.Tensor:
    """torchaudio `_apply_sinc_resample_kernel` on [B, S] -> [B, ceil(new*S/orig)]."""
    B, S = wave.shape
    x = F.pad(wave, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)           # [B, new, frames]
    y = y.transpose(1, 2).reshape(B, -1)
    return y[:, : int(math.ceil(new * S / orig))]


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int, device="cpu") -> torch.Tensor:
    """[n_freqs, n_mels] triangular filters, HTK mel scale, norm None (torchaudio.functional.melscale_fbanks)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, device=device)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, device=device)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def _db_topdb_crop(mel: torch.Tensor, n_frames: int) -> torch.Tensor:
    """mel power [B, 96, T] -> dB with the batch-wide top_db floor, centre crop / -80 pad to n_frames (:195,208-221)."""
    mel = mel.clamp_min(1e-8)
    mel_db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    # top_db over the whole batch tensor (see module doc).  The floor is NOT detached: torchaudio's amplitude_to_DB takes
    # torch.max(x_db, x_db.amax() - top_db) with the graph attached, so every bin clamped to the floor sends its upstream
    # gradient to the peak bin (checked by a finite-difference CPU test)
    mel_db = torch.max(mel_db, (mel_db.amax() - 80.0))
    t = int(mel_db.shape[-1])
    if t > n_frames:
        start = (t - n_frames) // 2
        mel_db = mel_db[:, :, start:start + n_frames]
    elif t < n_frames:
        mel_db = torch.cat([mel_db, mel_db.new_full((mel_db.shape[0], mel_db.shape[1], n_frames - t), -80.0)], dim=2)
    return mel_db


def wave_to_logmel_torch(wav_b_1_t: torch.Tensor, codec_sr: int, lab1_sr: int, n_frames: int) -> torch.Tensor:
    """The adapter in plain torch ops on whatever device the waveform lives on (conv1d, torch.stft, matmul): the specification."""
    dev = wav_b_1_t.device
    y = wav_b_1_t.squeeze(1).float()
    if codec_sr != lab1_sr:
        k, width, orig, new = sinc_resample_kernel(codec_sr, lab1_sr, device=dev)
        y = resample(y, k, width, orig, new)
    n_fft, hop, n_mels = 1024, 256, 96
    window = torch.hann_window(n_fft, periodic=True, device=dev)
    fb = melscale_fbanks(n_fft // 2 + 1, 20.0, float(lab1_sr) * 0.5, n_mels, lab1_sr, device=dev)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                            # [B, 513, frames]
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return _db_topdb_crop(mel, n_frames)


class Lab1WaveAdapter:
    """Same constructor and call as the reference's `_Lab1WaveAdapter` (:179-222): wav [B, 1, S] -> log-mel dB [B, 96, n_frames].
    GPU only (the contractions are HIP kernels); `wave_to_logmel_torch` is the torch-op form."""

    def __init__(self, codec_sr: int, lab1_sr: int, n_frames: int, device: torch.device):
        from . import _lib as L
        from . import ops
        self.codec_sr, self.lab1_sr, self.n_frames, self.device = int(codec_sr), int(lab1_sr), int(n_frames), torch.device(device)
        self.n_fft, self.hop, self.n_mels = 1024, 256, 96
        self._L, self._ops = L, ops
        self._packs = None

    def _build(self, device):
        """Packed filter banks, once (frozen): the polyphase kernels as a 2-row-tap conv over rows of `orig` samples, the mel
        filterbank as a k = 1 conv over the zero-padded power rows."""
        ops, L = self._ops, self._L
        packs = {}
        if self.codec_sr != self.lab1_sr:
            k, width, orig, new = sinc_resample_kernel(self.codec_sr, self.lab1_sr, device="cpu")
            taps = k.shape[-1]                                   # 2 * width + orig
            jrows = (taps + orig - 1) // orig                    # row-taps (2 for 24 kHz -> 22.05 kHz)
            cout = (new + 7) // 8 * 8
            cin = orig
            if cin % 8:
                raise ValueError(f"resample ratio {orig}:{new}: the row width {orig} must be a multiple of 8")
            w = torch.zeros((cout, jrows * orig), dtype=torch.float32)
            w[:new, :taps] = k[:, 0, :]
            w = w.view(cout, jrows, orig).permute(0, 2, 1).contiguous()  # [cout][cin][k]
            g = ops.ConvGeom("conv", cin, cout, jrows, 1, 0, 1)
            packs["rs"] = (g, *ops.pack_frozen(g, w.to(device), None, torch.float32), torch.zeros(cout, device=device), width, orig, new, jrows)
        nb, M = ops.stft_bins(self.n_fft)
        ldp = (nb + 7) // 8 * 8
        fb = melscale_fbanks(nb, 20.0, float(self.lab1_sr) * 0.5, self.n_mels, self.lab1_sr, device="cpu")  # [nb, 96]
        wm = torch.zeros((self.n_mels, ldp, 1), dtype=torch.float32)
        wm[:, :nb, 0] = fb.t()
        gm = ops.ConvGeom("conv", ldp, self.n_mels, 1, 1, 0, 1)
        packs["mel"] = (gm, *ops.pack_frozen(gm, wm.to(device), None, torch.float32), torch.zeros(self.n_mels, device=device), nb, ldp)
        self._packs = packs

    def __call__(self, wav_b_1_t: torch.Tensor) -> torch.Tensor:
        ops, L = self._ops, self._L
        L.require_cuda(wav_b_1_t)
        if self._packs is None:
            self._build(wav_b_1_t.device)
        y = wav_b_1_t.squeeze(1).float()
        B, S = y.shape
        if "rs" in self._packs:
            g, wp_f, wp_d, bias, width, orig, new, jrows = self._packs["rs"]
            frames = S // orig + 1
            rows = frames + jrows - 1
            xp = torch.nn.functional.pad(y, (width, rows * orig - width - S)).view(B, rows, orig)   # zero padding (width, width + orig)
            out = ops.ConvFrozen.apply(xp, None, wp_f, wp_d, bias, g, L.ACT_NONE, False, False)       # [B, frames, cout]
            y = out[:, :, :new].reshape(B, frames * new)[:, : int(math.ceil(new * S / orig))]
        spec = ops.StftSpec.apply(y.contiguous(), self.n_fft, self.hop, self.n_fft, True)              # [B, F, M] (re | im)
        gm, wm_f, wm_d, bias_m, nb, ldp = self._packs["mel"]
        power = ops.PowerSpec.apply(spec, nb, ldp)                                                     # [B, F, ldp]
        mel = ops.ConvFrozen.apply(power, None, wm_f, wm_d, bias_m, gm, L.ACT_NONE, False, False)      # [B, F, 96]
        return _db_topdb_crop(mel.permute(0, 2, 1), self.n_frames)
