"""Mel-track models of the reference (lab 3/src/lab3_models.py) with their convolutions on the HIP gathered-GEMM kernels.

SURVEY.md section 8(f) next-4, first slice.  Same class names, constructor arguments, forward signatures and state_dict keys as
the reference (`ReconstructionDecoder` :37-127, `MRFResidualBlock` :10-34, `MelDiscriminator` :130-191 and its MultiScale / SubBand /
MultiPeriod / Hybrid wrappers :194-408, `StyleCritic` :411-427, `bce_logits` :430-438), so checkpoints load either way.

How a 2-D convolution runs here (gan_amd/conv2d.py): activations are channels-last [B, H, W, C]; the kh kernel ROWS are gathered
along H into the channel axis (one strided-slice concatenation, kh / stride_h times the input), and what remains -- a 1-D
convolution along W with kh*C input channels -- is the library's implicit GEMM (`ga_conv` forward / data gradient, `ga_wgrad`), bias
and LeakyReLU in its epilogue where they follow directly.  ConvTranspose2d(k4, s2, p1) is the same 3x3 implicit GEMM onto 4x the
channels (each output phase uses a 2x2 subset of the 4x4 kernel) followed by the pixel shuffle.  The parameters live in ordinary
torch layers (that is what keeps the keys and the initialisation identical); normalisation layers, activations between them, the
pooled heads and the Linear layers run as the torch modules they are (which parts of the mel track are
HIP and which are not).  spectral_norm=True wraps the same layers the reference wraps (:77-78, :152-153) with torch's own
`nn.utils.spectral_norm`, so the `weight_orig` / `weight_u` / `weight_v` keys, the one power iteration per training-mode forward and
the gradient through sigma are torch's; the normalised weight it produces is what the HIP convolution packs and multiplies.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from torch.nn.utils.spectral_norm import SpectralNorm

from . import _lib as L
from .conv2d import conv2d_nhwc, conv_transpose2d_k4s2p1_nhwc, norm2d_nhwc, norm2d_supported, to_nchw_view, to_nhwc


def _maybe_sn(layer: nn.Module, flag: bool) -> nn.Module:
    return nn.utils.spectral_norm(layer) if bool(flag) else layer


def _weight_of(m: nn.Module) -> torch.Tensor:
    """The layer's weight as its own forward would see it: a spectral-norm layer recomputes `weight` in a forward pre-hook (power
    iteration in training mode, then weight_orig / sigma), which has to be run by hand because the layer's forward is not called."""
    for hook in m._forward_pre_hooks.values():
        if isinstance(hook, SpectralNorm):
            hook(m, None)
    return m.weight


def _dtype(precision: str) -> torch.dtype:
    if precision not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    return torch.float32 if precision == "fp32" else torch.bfloat16


def _run(layers: Sequence[nn.Module], h: torch.Tensor) -> torch.Tensor:
    """A reference nn.Sequential on an NCHW-shaped, channels-last-strided tensor: convolutions through the HIP path (with a
    directly following LeakyReLU(0.2) fused into the epilogue), everything else as the torch module it is."""
    i, n = 0, len(layers)
    while i < n:
        m = layers[i]
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            fuse = i + 1 < n and isinstance(layers[i + 1], nn.LeakyReLU) and abs(layers[i + 1].negative_slope - 0.2) < 1e-12
            act = L.ACT_LRELU if fuse else L.ACT_NONE
            x = to_nhwc(h)
            if isinstance(m, nn.ConvTranspose2d):
                if tuple(m.kernel_size) != (4, 4) or tuple(m.stride) != (2, 2) or tuple(m.padding) != (1, 1):
                    raise NotImplementedError("only ConvTranspose2d(kernel 4, stride 2, padding 1) is mapped (the decoder's up-blocks)")
                y = conv_transpose2d_k4s2p1_nhwc(x, _weight_of(m), m.bias, act)
            else:
                if m.stride[0] != m.stride[1] or m.padding[0] != m.padding[1] or tuple(m.dilation) != (1, 1) or m.groups != 1:
                    raise NotImplementedError("square stride / padding, no dilation, no groups")
                y = conv2d_nhwc(x, _weight_of(m), m.bias, int(m.stride[0]), int(m.padding[0]), act)
            h = to_nchw_view(y)
            i += 2 if fuse else 1
            continue
        if isinstance(m, (nn.InstanceNorm2d, nn.BatchNorm2d)) and norm2d_supported(m, h):
            fuse = i + 1 < n and isinstance(layers[i + 1], nn.LeakyReLU) and abs(layers[i + 1].negative_slope - 0.2) < 1e-12
            h = to_nchw_view(norm2d_nhwc(m, to_nhwc(h), L.ACT_LRELU if fuse else L.ACT_NONE))
            i += 2 if fuse else 1
            continue
        if isinstance(m, nn.PixelShuffle):
            h = F.pixel_shuffle(h, m.upscale_factor)
        else:
            h = m(h)
        i += 1
    return h


class MRFResidualBlock(nn.Module):
    """reference lab3_models.py:10-34: mean of k x k -> LeakyReLU -> 3 x 3 branches, a 1 x 1 conv, residual."""

    def __init__(self, channels: int, kernels: Sequence[int] = (3, 7, 11), dilation: int = 1):
        super().__init__()
        if int(dilation) != 1:
            raise NotImplementedError("dilation 1 only (the decoder never passes another)")
        self.branches = nn.ModuleList()
        for k in kernels:
            self.branches.append(nn.Sequential(
                nn.Conv2d(channels, channels, kernel_size=int(k), stride=1, padding=int(k) // 2),
                nn.LeakyReLU(0.2, inplace=True),
                nn.Conv2d(channels, channels, kernel_size=3, stride=1, padding=1)))
        self.out = nn.Conv2d(channels, channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = None
        for b in self.branches:
            t = _run(list(b), x)
            y = t if y is None else y + t
        y = y / float(len(self.branches))
        return x + _run([self.out], y)


class ReconstructionDecoder(nn.Module):
    """reference lab3_models.py:37-127: [B, zc_dim] + [B, cond_dim] -> mel_norm [B, n_mels, n_frames] in [-1, 1].
    Extra keyword: precision="fp32"|"bf16" (bf16 = bf16 MFMA operands, fp32 accumulation, fp32 parameters)."""

    def __init__(self, zc_dim: int = 128, cond_dim: int = 160, n_mels: int = 96, n_frames: int = 256, base_channels: int = 256,
                 norm: str = "instance", upsample: str = "transpose", spectral_norm: bool = False, mrf: bool = False,
                 mrf_kernels: Sequence[int] = (3, 7, 11), precision: str = "fp32"):
        super().__init__()
        if n_mels % 8 != 0 or n_frames % 8 != 0:
            raise ValueError("n_mels and n_frames must be divisible by 8.")
        sn = bool(spectral_norm)
        self.n_mels, self.n_frames = int(n_mels), int(n_frames)
        self.h0, self.w0 = self.n_mels // 8, self.n_frames // 8
        self.base = int(base_channels)
        up_key = str(upsample).strip().lower()
        if up_key not in {"transpose", "pixelshuffle", "nearest"}:
            raise ValueError("upsample must be one of: {'transpose', 'pixelshuffle', 'nearest'}")
        norm_key = str(norm).strip().lower()
        if norm_key not in {"instance", "batch"}:
            raise ValueError("norm must be one of: {'instance', 'batch'}")
        self.precision = precision
        self._dt = _dtype(precision)

        def norm2d(c):
            return nn.InstanceNorm2d(c, affine=True) if norm_key == "instance" else nn.BatchNorm2d(c)

        def up_block(cin, cout):  # the layer ORDER fixes the state_dict indices (up1.0 / up1.1 / up1.2 ...)
            if up_key == "transpose":
                return nn.Sequential(_maybe_sn(nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1), sn), norm2d(cout),
                                     nn.LeakyReLU(0.2, inplace=True))
            if up_key == "nearest":
                return nn.Sequential(nn.Upsample(scale_factor=2, mode="nearest"), _maybe_sn(nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1), sn),
                                     norm2d(cout), nn.LeakyReLU(0.2, inplace=True))
            return nn.Sequential(_maybe_sn(nn.Conv2d(cin, cout * 4, kernel_size=3, stride=1, padding=1), sn), nn.PixelShuffle(2), norm2d(cout),
                                 nn.LeakyReLU(0.2, inplace=True))

        b = self.base
        self.fc = _maybe_sn(nn.Linear(int(zc_dim + cond_dim), b * self.h0 * self.w0), sn)
        self.up1, self.up2, self.up3 = up_block(b, b // 2), up_block(b // 2, b // 4), up_block(b // 4, b // 8)
        self.mrf1 = MRFResidualBlock(b // 2, kernels=mrf_kernels) if mrf else nn.Identity()
        self.mrf2 = MRFResidualBlock(b // 4, kernels=mrf_kernels) if mrf else nn.Identity()
        self.mrf3 = MRFResidualBlock(b // 8, kernels=mrf_kernels) if mrf else nn.Identity()
        self.out = _maybe_sn(nn.Conv2d(b // 8, 1, kernel_size=3, stride=1, padding=1), sn)

    def forward(self, z_content: torch.Tensor, v_target: torch.Tensor) -> torch.Tensor:
        L.require_cuda(z_content, v_target)
        x = torch.cat([z_content, v_target], dim=1)
        h = self.fc(x).view(x.shape[0], self.base, self.h0, self.w0).to(self._dt)
        h = h.contiguous(memory_format=torch.channels_last)
        for up, mrf in ((self.up1, self.mrf1), (self.up2, self.mrf2), (self.up3, self.mrf3)):
            h = _run(list(up), h)
            h = mrf(h)
        return torch.tanh(_run([self.out], h).float()).squeeze(1)


class MelDiscriminator(nn.Module):
    """reference lab3_models.py:130-191: four stride-2 4 x 4 convs (BatchNorm from the second on), global average pool, linear head,
    optional projection term <normalize(h), normalize(cond_proj(cond))>.  Returns logits [B] (and the pooled features)."""

    def __init__(self, in_channels: int = 1, base_channels: int = 64, cond_dim: int = 0, spectral_norm: bool = False, precision: str = "fp32"):
        super().__init__()
        sn = bool(spectral_norm)
        c = int(base_channels)
        self.cond_dim = int(cond_dim)
        self.precision = precision
        self._dt = _dtype(precision)
        layers = [_maybe_sn(nn.Conv2d(in_channels, c, kernel_size=4, stride=2, padding=1), sn), nn.LeakyReLU(0.2, inplace=True)]
        for k in (1, 2, 4):
            layers += [_maybe_sn(nn.Conv2d(c * k, c * k * 2, kernel_size=4, stride=2, padding=1), sn), nn.BatchNorm2d(c * k * 2),
                       nn.LeakyReLU(0.2, inplace=True)]
        layers.append(nn.AdaptiveAvgPool2d((1, 1)))
        self.net = nn.Sequential(*layers)
        self.head = _maybe_sn(nn.Linear(c * 8, 1), sn)
        self.cond_proj = _maybe_sn(nn.Linear(self.cond_dim, c * 8), sn) if self.cond_dim > 0 else None

    def forward(self, mel_norm: torch.Tensor, cond: Optional[torch.Tensor] = None, return_features: bool = False):
        L.require_cuda(mel_norm)
        x = mel_norm.unsqueeze(1).to(self._dt).contiguous(memory_format=torch.channels_last)
        h = _run(list(self.net), x).flatten(1).float()
        logit = self.head(h).squeeze(1)
        if self.cond_proj is not None and cond is not None:
            logit = logit + torch.sum(F.normalize(h, dim=-1) * F.normalize(self.cond_proj(cond), dim=-1), dim=-1)
        return (logit, h) if return_features else logit


def _mean_logit_cat_feats(outs, return_features: bool):
    logit = torch.stack([o[0] for o in outs], dim=0).mean(dim=0)
    return (logit, torch.cat([o[1] for o in outs], dim=-1)) if return_features else logit


class MultiScaleMelDiscriminator(nn.Module):
    """reference lab3_models.py:194-251: the same discriminator on the mel at full, 1/2 and 1/4 resolution (2 x 2 average pooling);
    mean logit, concatenated features."""

    def __init__(self, in_channels: int = 1, base_channels: int = 64, cond_dim: int = 0, num_scales: int = 3, spectral_norm: bool = False,
                 precision: str = "fp32"):
        super().__init__()
        if int(num_scales) < 1:
            raise ValueError("num_scales must be >= 1")
        self.num_scales = int(num_scales)
        self.discriminators = nn.ModuleList([MelDiscriminator(in_channels, base_channels, cond_dim, spectral_norm, precision)
                                             for _ in range(self.num_scales)])

    def forward(self, mel_norm, cond=None, return_features: bool = False):
        x, outs = mel_norm, []
        for i, d in enumerate(self.discriminators):
            outs.append(d(x, cond, return_features=True))
            if i + 1 < self.num_scales:
                x = F.avg_pool2d(x.unsqueeze(1), kernel_size=(2, 2), stride=(2, 2)).squeeze(1)
        return _mean_logit_cat_feats(outs, return_features)


class SubBandMelDiscriminator(nn.Module):
    """reference lab3_models.py:254-322: low / mid / high mel bands judged by three discriminators."""

    def __init__(self, in_channels: int = 1, base_channels: int = 64, cond_dim: int = 0, low_bins: int = 32, mid_bins: int = 32,
                 spectral_norm: bool = False, precision: str = "fp32"):
        super().__init__()
        self.low_bins, self.mid_bins = int(low_bins), int(mid_bins)
        self.d_low = MelDiscriminator(in_channels, base_channels, cond_dim, spectral_norm, precision)
        self.d_mid = MelDiscriminator(in_channels, base_channels, cond_dim, spectral_norm, precision)
        self.d_high = MelDiscriminator(in_channels, base_channels, cond_dim, spectral_norm, precision)

    def _split(self, mel_norm: torch.Tensor):
        n = int(mel_norm.shape[1])
        low = max(8, min(self.low_bins, n - 16))
        hi = low + max(8, min(self.mid_bins, n - low - 8))
        if hi >= n:
            hi = max(8, n - 8)
            low = max(8, min(low, hi - 8))
        return mel_norm[:, :low, :], mel_norm[:, low:hi, :], mel_norm[:, hi:, :]

    def forward(self, mel_norm, cond=None, return_features: bool = False):
        outs = [d(x, cond, return_features=True) for d, x in zip((self.d_low, self.d_mid, self.d_high), self._split(mel_norm))]
        return _mean_logit_cat_feats(outs, return_features)


class MultiPeriodMelDiscriminator(nn.Module):
    """reference lab3_models.py:325-375: every branch sees the mel at another temporal stride (views shorter than 16 frames fall back
    to the full mel)."""

    def __init__(self, in_channels: int = 1, base_channels: int = 64, cond_dim: int = 0, periods: tuple = (1, 2, 3, 5),
                 spectral_norm: bool = False, precision: str = "fp32"):
        super().__init__()
        self.periods = tuple(int(max(1, p)) for p in periods)
        self.discriminators = nn.ModuleList([MelDiscriminator(in_channels, base_channels, cond_dim, spectral_norm, precision)
                                             for _ in self.periods])

    def forward(self, mel_norm, cond=None, return_features: bool = False):
        outs = []
        for p, d in zip(self.periods, self.discriminators):
            x = mel_norm[:, :, ::p]
            outs.append(d(x if x.shape[-1] >= 16 else mel_norm, cond, return_features=True))
        return _mean_logit_cat_feats(outs, return_features)


class HybridMelDiscriminator(nn.Module):
    """reference lab3_models.py:378-408: mean of the multi-scale and the multi-period critic."""

    def __init__(self, cond_dim: int = 0, num_scales: int = 3, periods: tuple = (1, 2, 3, 5), spectral_norm: bool = False,
                 precision: str = "fp32"):
        super().__init__()
        self.ms = MultiScaleMelDiscriminator(cond_dim=cond_dim, num_scales=num_scales, spectral_norm=spectral_norm, precision=precision)
        self.mp = MultiPeriodMelDiscriminator(cond_dim=cond_dim, periods=periods, spectral_norm=spectral_norm, precision=precision)

    def forward(self, mel_norm, cond=None, return_features: bool = False):
        l_ms, f_ms = self.ms(mel_norm, cond, return_features=True)
        l_mp, f_mp = self.mp(mel_norm, cond, return_features=True)
        logit = 0.5 * (l_ms + l_mp)
        return (logit, torch.cat([f_ms, f_mp], dim=-1)) if return_features else logit


class StyleCritic(nn.Module):
    """reference lab3_models.py:411-427: a two-layer MLP on frozen-encoder z_style features (plain torch: 2 x [B, 256] GEMMs)."""

    def __init__(self, in_dim: int, n_classes: int, hidden_dim: int = 256, dropout: float = 0.1):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(int(in_dim), int(hidden_dim)), nn.LeakyReLU(0.2, inplace=True), nn.Dropout(float(dropout)),
                                 nn.Linear(int(hidden_dim), int(n_classes)))

    def forward(self, z_style: torch.Tensor) -> torch.Tensor:
        return self.net(z_style)


def bce_logits(logits: torch.Tensor, target_is_real: bool, real_label: float = 1.0, fake_label: float = 0.0) -> torch.Tensor:
    """reference lab3_models.py:430-438."""
    t = torch.full_like(logits, float(real_label) if target_is_real else float(fake_label))
    return F.binary_cross_entropy_with_logits(logits, t)
