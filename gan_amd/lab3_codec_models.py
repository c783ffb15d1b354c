"""MI355X-native drop-in for the reference's `lab 3/src/lab3_codec_models.py`.

Same class / function names, constructor and forward signatures, return structures and `state_dict()`
keys as the reference (cited per symbol); the arithmetic runs in the HIP kernels behind gan_amd.ops.
Tensors cross the API as [B, C, T] like the reference; inside, activations are channels-last [B, T, C]
(the returned feature maps are [B, C, T]-shaped views of those buffers).  GPU only.
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .ops import ConvGeom

_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16}


def _act_dtype(precision: str) -> torch.dtype:
    if precision not in _DTYPES:
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
    return _DTYPES[precision]


def _to_nlc(x_bct: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[B, C, T] (any strides) -> contiguous [B, T, C] in the activation dtype (free when it already is)."""
    return x_bct.permute(0, 2, 1).to(dtype).contiguous()


class FiLM(nn.Module):
    """reference lab3_codec_models.py:11-25.  The 2-layer MLP on the [B, cond] vector (0.4 MMAC) is a plain
    library GEMM; the broadcast x*(1+a)+b is fused into the GroupNorm kernel (ops.GnFilmAct)."""

    def __init__(self, cond_dim: int, n_channels: int):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(cond_dim, n_channels * 2), nn.SiLU(), nn.Linear(n_channels * 2, n_channels * 2))

    def ab(self, cond: torch.Tensor) -> torch.Tensor:
        return self.proj(cond)


class ResidualFiLMBlock(nn.Module):
    """reference lab3_codec_models.py:28-47 (parameter containers are the same torch modules => same keys/init)."""

    def __init__(self, channels: int, cond_dim: int, dilation: int):
        super().__init__()
        self.conv1 = nn.Conv1d(channels, channels, kernel_size=3, padding=dilation, dilation=dilation)
        self.conv2 = nn.Conv1d(channels, channels, kernel_size=3, padding=1)
        self.norm1 = nn.GroupNorm(num_groups=8, num_channels=channels)
        self.norm2 = nn.GroupNorm(num_groups=8, num_channels=channels)
        self.film1 = FiLM(cond_dim=cond_dim, n_channels=channels)
        self.film2 = FiLM(cond_dim=cond_dim, n_channels=channels)
        self.g1 = ConvGeom("conv", channels, channels, 3, 1, dilation, dilation)
        self.g2 = ConvGeom("conv", channels, channels, 3, 1, 1, 1)


def _conv_train(x, conv: nn.Conv1d, geom: ConvGeom, dtype, post_act=L.ACT_NONE, weight_g=None, weight_v=None):
    v = conv.weight if weight_v is None else weight_v
    wp_f, wp_d, carrier = ops.PackTrainable.apply(v, weight_g, geom, dtype)
    return ops.ConvTrain.apply(x, wp_f, wp_d, carrier, conv.bias, geom, post_act, v if weight_g is None else None)


class CodecLatentTranslator(nn.Module):
    """reference lab3_codec_models.py:50-119: FiLM-conditioned dilated Conv1d ResNet on codec latents."""

    def __init__(self, in_channels: int = 128, z_content_dim: int = 128, z_style_dim: int = 128, hidden_channels: int = 256,
                 n_blocks: int = 10, noise_dim: int = 32, residual_scale: float = 0.5, direct_output: bool = False,
                 precision: str = "fp32"):
        super().__init__()
        self.in_channels = int(in_channels)
        self.z_content_dim = int(z_content_dim)
        self.z_style_dim = int(z_style_dim)
        self.noise_dim = int(noise_dim)
        self.residual_scale = float(residual_scale)
        self.direct_output = bool(direct_output)
        self.precision = precision
        cond_dim = self.z_content_dim + self.z_style_dim + self.noise_dim
        self.in_proj = nn.Conv1d(self.in_channels, hidden_channels, kernel_size=3, padding=1)
        dilations = [1, 2, 4, 8, 1, 2, 4, 8, 1, 2]
        if int(n_blocks) != len(dilations):
            dilations = [2 ** (i % 4) for i in range(int(n_blocks))]
        self.blocks = nn.ModuleList([ResidualFiLMBlock(hidden_channels, cond_dim, int(d)) for d in dilations])
        self.out_norm = nn.GroupNorm(num_groups=8, num_channels=hidden_channels)
        self.out_proj = nn.Conv1d(hidden_channels, self.in_channels, kernel_size=3, padding=1)
        self._g_in = ConvGeom("conv", self.in_channels, hidden_channels, 3, 1, 1, 1)
        self._g_out = ConvGeom("conv", hidden_channels, self.in_channels, 3, 1, 1, 1)

    def sample_noise(self, batch_size: int, device: torch.device) -> torch.Tensor:
        return torch.randn(batch_size, self.noise_dim, device=device)

    def forward(self, q_src: torch.Tensor, z_content: torch.Tensor, z_style_tgt: torch.Tensor,
                noise: Optional[torch.Tensor] = None, style_dropout_p: float = 0.0, style_jitter_std: float = 0.0) -> torch.Tensor:
        L.require_cuda(q_src)
        dtype = _act_dtype(self.precision)
        b = int(q_src.shape[0])
        if noise is None:
            noise = torch.zeros((b, self.noise_dim), device=q_src.device, dtype=torch.float32)
        zs = z_style_tgt
        if style_dropout_p > 0.0:
            keep = (torch.rand((b, 1), device=q_src.device) >= float(style_dropout_p)).to(zs.dtype)
            zs = zs * keep
        if style_jitter_std > 0.0:
            zs = zs + torch.randn_like(zs) * float(style_jitter_std)
        cond = torch.cat([z_content, zs, noise], dim=1).float()

        # All 2*n_blocks FiLM MLPs read the same cond vector: their first Linear layers are ONE GEMM on the concatenated
        # weights and their second layers one batched GEMM, instead of 4*n_blocks tiny library launches (and as many again
        # in the backward pass).  Same arithmetic per output element; autograd routes the gradients back to each module.
        films = [f for blk in self.blocks for f in (blk.film1, blk.film2)]
        n_f = len(films)
        hid = F.silu(F.linear(cond, torch.cat([f.proj[0].weight for f in films], 0), torch.cat([f.proj[0].bias for f in films], 0)))
        hid = hid.view(b, n_f, -1).transpose(0, 1)                                   # [n_f, B, 2C]
        w2 = torch.stack([f.proj[2].weight for f in films], 0).transpose(1, 2)       # [n_f, 2C(in), 2C(out)]
        b2 = torch.stack([f.proj[2].bias for f in films], 0).unsqueeze(1)            # [n_f, 1, 2C]
        abs_ = torch.baddbmm(b2, hid, w2).unbind(0)                                  # n_f x [B, 2C], contiguous

        q = _to_nlc(q_src, dtype)
        ops.prepack_layers([(self.in_proj.weight, None, self._g_in)]
                           + [(c.weight, None, g) for blk in self.blocks for c, g in ((blk.conv1, blk.g1), (blk.conv2, blk.g2))]
                           + [(self.out_proj.weight, None, self._g_out)], dtype)
        h = _conv_train(q, self.in_proj, self._g_in, dtype)
        for i, blk in enumerate(self.blocks):
            r = _conv_train(h, blk.conv1, blk.g1, dtype)
            r = ops.GnFilmAct.apply(r, blk.norm1.weight, blk.norm1.bias, abs_[2 * i], None, 8, L.ACT_SILU, blk.norm1.eps)
            r = _conv_train(r, blk.conv2, blk.g2, dtype)
            h = ops.GnFilmAct.apply(r, blk.norm2.weight, blk.norm2.bias, abs_[2 * i + 1], h, 8, L.ACT_SILU, blk.norm2.eps)
        h = ops.GnFilmAct.apply(h, self.out_norm.weight, self.out_norm.bias, None, None, 8, L.ACT_SILU, self.out_norm.eps)
        raw = _conv_train(h, self.out_proj, self._g_out, dtype)
        if self.direct_output:
            return raw.permute(0, 2, 1)
        return ops.TanhResidual.apply(raw, q, self.residual_scale).permute(0, 2, 1)


class WaveDiscBlock(nn.Module):
    """reference lab3_codec_models.py:122-128: weight_norm(Conv1d) + leaky_relu(0.2), fused into one kernel epilogue."""

    def __init__(self, in_ch: int, out_ch: int, k: int, s: int, p: int):
        super().__init__()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.conv = nn.utils.weight_norm(nn.Conv1d(in_ch, out_ch, kernel_size=k, stride=s, padding=p))
        self.geom = ConvGeom("conv", in_ch, out_ch, k, s, p, 1)

    def run(self, x_nlc: torch.Tensor, act: int) -> torch.Tensor:
        return _conv_train(x_nlc, self.conv, self.geom, x_nlc.dtype, act, self.conv.weight_g, self.conv.weight_v)


FUSED_DISC_CHAIN = True  # tests switch this off to compare against the layer-by-layer autograd nodes


class SingleScaleWaveDiscriminator(nn.Module):
    """reference lab3_codec_models.py:131-152"""

    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([
            WaveDiscBlock(1, 32, 15, 1, 7),
            WaveDiscBlock(32, 64, 41, 4, 20),
            WaveDiscBlock(64, 128, 41, 4, 20),
            WaveDiscBlock(128, 256, 41, 4, 20),
            WaveDiscBlock(256, 512, 41, 4, 20),
        ])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.out = nn.utils.weight_norm(nn.Conv1d(512, 1, kernel_size=3, stride=1, padding=1))
        self._g_out = ConvGeom("conv", 512, 1, 3, 1, 1, 1)

    def forward_nlc(self, x: torch.Tensor) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        if not FUSED_DISC_CHAIN:
            feats: List[torch.Tensor] = []
            h = x
            for blk in self.blocks:
                h = blk.run(h, L.ACT_LRELU)
                feats.append(h)
            logits = _conv_train(h, self.out, self._g_out, h.dtype, L.ACT_NONE, self.out.weight_g, self.out.weight_v)
            return logits, feats
        # one autograd node for the whole scale (ops.DiscChain): its backward fuses masks and gradient sums across layers
        convs = [(blk.conv, blk.geom) for blk in self.blocks] + [(self.out, self._g_out)]
        flat = []
        for conv, geom in convs:
            wp_f, wp_d, carrier = ops.PackTrainable.apply(conv.weight_v, conv.weight_g, geom, x.dtype)
            flat += [wp_f, wp_d, carrier, conv.bias]
        outs = ops.DiscChain.apply(x, tuple(g for _, g in convs), *flat)
        return outs[-1], list(outs[:-1])

    def forward_real_fake_nlc(self, x_real: torch.Tensor, x_fake: torch.Tensor, want_fm: bool = False):
        """G-step pair of passes as one (ops.DiscChainSplit): ((logits, feats) of x_real without gradient, the same of x_fake
        [, this scale's feature-matching term when want_fm])."""
        convs = [(blk.conv, blk.geom) for blk in self.blocks] + [(self.out, self._g_out)]
        flat = []
        with torch.no_grad():
            for conv, geom in convs:
                wp_f, wp_d, _ = ops.PackTrainable.apply(conv.weight_v, conv.weight_g, geom, x_fake.dtype)
                flat += [wp_f, wp_d, conv.bias]
        outs = ops.DiscChainSplit.apply(x_real, x_fake, tuple(g for _, g in convs), bool(want_fm), *flat)
        n = len(convs)
        pair = (outs[n - 1], list(outs[:n - 1])), (outs[2 * n - 1], list(outs[n:2 * n - 1]))
        return pair + (outs[2 * n],) if want_fm else pair

    def forward(self, wav: torch.Tensor) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        logits, feats = self.forward_nlc(wav.permute(0, 2, 1).contiguous())
        return logits.permute(0, 2, 1), [f.permute(0, 2, 1) for f in feats]


class MultiScaleWaveDiscriminator(nn.Module):
    """reference lab3_codec_models.py:155-168"""

    def __init__(self, n_scales: int = 3, precision: str = "fp32"):
        super().__init__()
        self.discs = nn.ModuleList([SingleScaleWaveDiscriminator() for _ in range(int(n_scales))])
        self.pool = nn.AvgPool1d(kernel_size=4, stride=2, padding=1)  # parameter-free; kept for attribute parity
        self.precision = precision

    def forward(self, wav: torch.Tensor) -> List[Tuple[torch.Tensor, List[torch.Tensor]]]:
        L.require_cuda(wav)
        dtype = _act_dtype(self.precision)
        if wav.dim() != 3 or wav.shape[1] != 1:
            raise ValueError(f"expected wav [B, 1, S], got {tuple(wav.shape)}")
        outs: List[Tuple[torch.Tensor, List[torch.Tensor]]] = []
        ops.prepack_layers(self._pack_layers(), dtype)
        x = wav.squeeze(1).to(dtype)  # [B, S] (a view: no select backward); with C == 1 the NLC and NCL layouts coincide
        for i, d in enumerate(self.discs):
            if i > 0:
                x = ops.AvgPool4s2.apply(x)
            logits, feats = d.forward_nlc(x.unsqueeze(-1))
            outs.append((logits.permute(0, 2, 1), [f.permute(0, 2, 1) for f in feats]))
        return outs

    def _pack_layers(self):
        return [(c.weight_v, c.weight_g, g) for d in self.discs
                for c, g in [(blk.conv, blk.geom) for blk in d.blocks] + [(d.out, d._g_out)]]

    def forward_real_fake(self, wav_real: torch.Tensor, wav_fake: torch.Tensor, want_fm: bool = False):
        """(D(wav_real) without gradient, D(wav_fake) with gradient to wav_fake only) for FROZEN discriminator weights --
        the generator step's two passes (reference lab3_codec_train.py:455-462) run as one pass over the joined batch.
        want_fm: a third result, multiscale_feature_matching_loss(real_outs, fake_outs), whose reductions run inside the
        same autograd nodes and whose gradient is formed in the data-gradient epilogues (the features returned for the
        fake half still carry gradient for any other use)."""
        L.require_cuda(wav_real, wav_fake)
        if any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("forward_real_fake needs the discriminator's parameters frozen (requires_grad False)")
        dtype = _act_dtype(self.precision)
        for w in (wav_real, wav_fake):
            if w.dim() != 3 or w.shape[1] != 1:
                raise ValueError(f"expected wav [B, 1, S], got {tuple(w.shape)}")
        if wav_real.shape[2] != wav_fake.shape[2]:
            raise ValueError("real and fake waveforms differ in length")
        ops.prepack_layers(self._pack_layers(), dtype)
        xr = wav_real.detach().squeeze(1).to(dtype)
        xf = wav_fake.squeeze(1).to(dtype)
        real_outs, fake_outs, fm_terms = [], [], []
        for i, d in enumerate(self.discs):
            if i > 0:
                with torch.no_grad():
                    xr = ops.AvgPool4s2.apply(xr)
                xf = ops.AvgPool4s2.apply(xf)
            res = d.forward_real_fake_nlc(xr.unsqueeze(-1), xf.unsqueeze(-1), want_fm)
            (lr, fr), (lf, ff) = res[0], res[1]
            if want_fm:
                fm_terms.append(res[2])
            real_outs.append((lr.permute(0, 2, 1), [f.permute(0, 2, 1) for f in fr]))
            fake_outs.append((lf.permute(0, 2, 1), [f.permute(0, 2, 1) for f in ff]))
        if want_fm:
            return real_outs, fake_outs, torch.stack(fm_terms).sum() / max(1, len(fm_terms))
        return real_outs, fake_outs


# ------------------------------------------------------------------------------------------------
# losses (reference lab3_codec_models.py:171-215); means are sums from the reduction kernels / numel
def hinge_d_loss(real_logits: torch.Tensor, fake_logits: torch.Tensor) -> torch.Tensor:
    return (ops.reduce_sum(real_logits, None, L.RED_RELU1M, 1.0 / real_logits.numel())
            + ops.reduce_sum(fake_logits, None, L.RED_RELU1P, 1.0 / fake_logits.numel()))


def hinge_g_loss(fake_logits: torch.Tensor) -> torch.Tensor:
    return ops.reduce_sum(fake_logits, None, L.RED_SUM, -1.0 / fake_logits.numel())


def feature_matching_loss(real_feats: Sequence[torch.Tensor], fake_feats: Sequence[torch.Tensor]) -> torch.Tensor:
    # F.l1_loss(a.detach(), b): the gradient goes to the fake features only; 1/numel is applied inside the reduction kernel
    terms = [ops.reduce_sum(b, a.detach(), L.RED_ABSDIFF, 1.0 / b.numel()) for a, b in zip(real_feats, fake_feats)]
    if not terms:
        # no feature maps, hence no device to put a result on: the reference returns the Python float 0.0 here (:181-186,
        # `loss = 0.0` divided by max(1, 0)); a 0-dim CPU zero behaves the same in the caller's weighted sum
        return torch.zeros(())
    return torch.stack(terms).sum() / max(1, len(fake_feats))


def multiscale_hinge_d_loss(real_outs, fake_outs) -> torch.Tensor:
    terms = [hinge_d_loss(real_logits, fake_logits) for (real_logits, _), (fake_logits, _) in zip(real_outs, fake_outs)]
    return torch.stack(terms).sum() / max(1, len(real_outs))


def multiscale_hinge_d_loss_joined(both_outs, b0: int) -> torch.Tensor:
    """multiscale_hinge_d_loss for the joined [real ; fake] D pass: `both_outs` as returned by the discriminator on the
    concatenated batch, whose first b0 items are the real ones."""
    if not 0 < b0 < int(both_outs[0][0].shape[0]):
        raise ValueError("b0 must split the batch into a real and a fake part")
    terms = [ops.HingeDJoined.apply(lg, int(b0)) for lg, _ in both_outs]
    return torch.stack(terms).sum() / max(1, len(both_outs))


def multiscale_hinge_g_loss(fake_outs) -> torch.Tensor:
    terms = [hinge_g_loss(fake_logits) for fake_logits, _ in fake_outs]
    return torch.stack(terms).sum() / max(1, len(fake_outs))


def multiscale_feature_matching_loss(real_outs, fake_outs) -> torch.Tensor:
    terms = [feature_matching_loss(rf, ff) for (_, rf), (_, ff) in zip(real_outs, fake_outs)]
    return torch.stack(terms).sum() / max(1, len(real_outs))


@dataclass
class CodecTrainWeights:
    """reference lab3_codec_models.py:218-229"""
    adv: float = 0.5
    latent_l1: float = 4.0
    latent_continuity: float = 1.0
    content: float = 2.0
    style: float = 3.0
    mrstft: float = 2.0
    feature_match: float = 1.0
    mode_seeking: float = 0.0
    style_push: float = 0.0
    delta_budget: float = 0.0
