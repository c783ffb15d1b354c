"""CPU restatement of the reference's MEL-TRACK GAN pieces -- TEST INFRASTRUCTURE ONLY.

Nothing under gan_amd/ may import this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

Functional torch-CPU (fp32) forms of
  * MRFResidualBlock, ReconstructionDecoder          /root/reference/lab 3/src/lab3_models.py:10-34, 37-127
  * MelDiscriminator and its four wrappers, bce_logits lab3_models.py:130-191, 194-408, 430-438
  * _gan_d_loss / _gan_g_loss                          /root/reference/lab 3/src/lab3_train.py:297-326
  * the body of one train_stage1 iteration             lab3_train.py:380-456 (D step, G step, Adam, optional clipping)
on plain state dicts with the reference's key names.  Pinned by tests/golden/melgan_golden.npz, which
tests/golden/make_golden_melgan.py generates from the reference's own classes (tests/test_mel_oracle_golden.py).
Spectral norm (a power-iteration buffer that changes on every forward) is not restated: the reference's default is off.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .codec_oracle import AdamWState, adamw_step, clip_grad_norm, mel_mrstft_loss

Tensor = torch.Tensor
SD = Dict[str, Tensor]

MEL_DB_MIN, MEL_DB_MAX = -80.0, 0.0  # lab3_bridge.py (denormalize_log_mel)


# ------------------------------------------------------------------------------------------------ decoder
@dataclass
class DecoderCfg:  # ReconstructionDecoder.__init__ (lab3_models.py:47-59)
    zc_dim: int = 128
    cond_dim: int = 160
    n_mels: int = 96
    n_frames: int = 256
    base_channels: int = 256
    norm: str = "instance"
    upsample: str = "transpose"
    mrf: bool = False
    mrf_kernels: Sequence[int] = (3, 7, 11)


def _norm2d(sd: SD, pre: str, x: Tensor, kind: str, buffers: Optional[SD], training: bool) -> Tensor:
    """nn.InstanceNorm2d(c, affine=True) (no running stats) or nn.BatchNorm2d(c) (lab3_models.py:80-83)."""
    if kind == "instance":
        return F.instance_norm(x, weight=sd[pre + ".weight"], bias=sd[pre + ".bias"], eps=1e-5)
    return _batch_norm(sd, pre, x, buffers, training)


def _batch_norm(sd: SD, pre: str, x: Tensor, buffers: Optional[SD], training: bool) -> Tensor:
    """nn.BatchNorm2d: batch statistics in training mode; the running buffers (momentum 0.1, unbiased variance) are updated in
    `buffers` when given, exactly as the module updates its own."""
    rm, rv = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, sd[pre + ".weight"], sd[pre + ".bias"], False, 0.1, 1e-5)
    y = F.batch_norm(x, None, None, sd[pre + ".weight"], sd[pre + ".bias"], True, 0.1, 1e-5)
    if buffers is not None:
        with torch.no_grad():
            xd = x.detach()
            n = xd.numel() // xd.shape[1]
            mean = xd.mean(dim=(0, 2, 3))
            var_u = xd.var(dim=(0, 2, 3), unbiased=True) if n > 1 else xd.var(dim=(0, 2, 3), unbiased=False)
            buffers[pre + ".running_mean"] = 0.9 * buffers.get(pre + ".running_mean", rm) + 0.1 * mean
            buffers[pre + ".running_var"] = 0.9 * buffers.get(pre + ".running_var", rv) + 0.1 * var_u
            nbt = buffers.get(pre + ".num_batches_tracked", sd.get(pre + ".num_batches_tracked"))
            if nbt is not None:
                buffers[pre + ".num_batches_tracked"] = nbt + 1
    return y


def _up_block(sd: SD, pre: str, x: Tensor, cfg: DecoderCfg, buffers, training) -> Tensor:
    """up_block (lab3_models.py:85-105): transpose | nearest | pixelshuffle, then norm and LeakyReLU(0.2)."""
    if cfg.upsample == "transpose":
        h = F.conv_transpose2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], stride=2, padding=1)
        ni = 1
    elif cfg.upsample == "nearest":
        h = F.interpolate(x, scale_factor=2, mode="nearest")
        h = F.conv2d(h, sd[pre + ".1.weight"], sd[pre + ".1.bias"], stride=1, padding=1)
        ni = 2
    else:
        h = F.conv2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], stride=1, padding=1)
        h = F.pixel_shuffle(h, 2)
        ni = 2
    h = _norm2d(sd, f"{pre}.{ni}", h, cfg.norm, buffers, training)
    return F.leaky_relu(h, 0.2)


def mrf_block(sd: SD, pre: str, x: Tensor, kernels: Sequence[int]) -> Tensor:
    """MRFResidualBlock.forward (lab3_models.py:29-34); dilation 1 (the decoder never passes another)."""
    y = 0.0
    for i, k in enumerate(kernels):
        h = F.conv2d(x, sd[f"{pre}.branches.{i}.0.weight"], sd[f"{pre}.branches.{i}.0.bias"], padding=int(k) // 2)
        h = F.leaky_relu(h, 0.2)
        y = y + F.conv2d(h, sd[f"{pre}.branches.{i}.2.weight"], sd[f"{pre}.branches.{i}.2.bias"], padding=1)
    y = y / float(len(kernels))
    return x + F.conv2d(y, sd[pre + ".out.weight"], sd[pre + ".out.bias"])


def decoder_forward(sd: SD, z_content: Tensor, v_target: Tensor, cfg: DecoderCfg = DecoderCfg(), buffers: Optional[SD] = None,
                    training: bool = True) -> Tensor:
    """ReconstructionDecoder.forward (lab3_models.py:116-127): [B, zc] + [B, cond] -> mel_norm [B, n_mels, n_frames] in [-1, 1]."""
    x = torch.cat([z_content, v_target], dim=1)
    h = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    h = h.view(x.shape[0], cfg.base_channels, cfg.n_mels // 8, cfg.n_frames // 8)
    for i in (1, 2, 3):
        h = _up_block(sd, f"up{i}", h, cfg, buffers, training)
        if cfg.mrf:
            h = mrf_block(sd, f"mrf{i}", h, cfg.mrf_kernels)
    return torch.tanh(F.conv2d(h, sd["out.weight"], sd["out.bias"], padding=1)).squeeze(1)


# ------------------------------------------------------------------------------------------------ discriminators
def mel_disc_forward(sd: SD, pre: str, mel_norm: Tensor, cond: Optional[Tensor], buffers: Optional[SD] = None,
                     training: bool = True) -> Tuple[Tensor, Tensor]:
    """MelDiscriminator.forward (lab3_models.py:176-191) -> (logit [B], pooled features [B, 8c]).  `pre` is the key prefix
    ("" or "discriminators.0." ...)."""
    h = mel_norm.unsqueeze(1)
    h = F.leaky_relu(F.conv2d(h, sd[pre + "net.0.weight"], sd[pre + "net.0.bias"], stride=2, padding=1), 0.2)
    for ci, bi in ((2, 3), (5, 6), (8, 9)):
        h = F.conv2d(h, sd[f"{pre}net.{ci}.weight"], sd[f"{pre}net.{ci}.bias"], stride=2, padding=1)
        h = F.leaky_relu(_batch_norm(sd, f"{pre}net.{bi}", h, buffers, training), 0.2)
    h = h.mean(dim=(2, 3))
    logit = F.linear(h, sd[pre + "head.weight"], sd[pre + "head.bias"]).squeeze(1)
    if (pre + "cond_proj.weight") in sd and cond is not None:
        c = F.normalize(F.linear(cond, sd[pre + "cond_proj.weight"], sd[pre + "cond_proj.bias"]), dim=-1)
        logit = logit + torch.sum(F.normalize(h, dim=-1) * c, dim=-1)
    return logit, h


def multiscale_forward(sd: SD, pre: str, mel_norm: Tensor, cond, num_scales: int = 3, buffers=None, training=True):
    """MultiScaleMelDiscriminator.forward (lab3_models.py:227-251)."""
    x, logits, feats = mel_norm, [], []
    for i in range(num_scales):
        li, fi = mel_disc_forward(sd, f"{pre}discriminators.{i}.", x, cond, buffers, training)
        logits.append(li)
        feats.append(fi)
        if i < num_scales - 1:
            x = F.avg_pool2d(x.unsqueeze(1), kernel_size=(2, 2), stride=(2, 2)).squeeze(1)
    return torch.stack(logits, dim=0).mean(dim=0), torch.cat(feats, dim=-1)


def subband_split(mel_norm: Tensor, low_bins: int = 32, mid_bins: int = 32):
    """SubBandMelDiscriminator._split (lab3_models.py:292-303)."""
    n_mels = int(mel_norm.shape[1])
    low = max(8, min(low_bins, n_mels - 16))
    mid = max(8, min(mid_bins, n_mels - low - 8))
    hi_start = low + mid
    if hi_start >= n_mels:
        hi_start = max(8, n_mels - 8)
        low = max(8, min(low, hi_start - 8))
    return mel_norm[:, :low, :], mel_norm[:, low:hi_start, :], mel_norm[:, hi_start:, :]


def subband_forward(sd: SD, pre: str, mel_norm: Tensor, cond, low_bins: int = 32, mid_bins: int = 32, buffers=None, training=True):
    """SubBandMelDiscriminator.forward (lab3_models.py:305-322)."""
    outs = [mel_disc_forward(sd, f"{pre}{name}.", x, cond, buffers, training)
            for name, x in zip(("d_low", "d_mid", "d_high"), subband_split(mel_norm, low_bins, mid_bins))]
    return torch.stack([o[0] for o in outs], dim=0).mean(dim=0), torch.cat([o[1] for o in outs], dim=-1)


def multiperiod_forward(sd: SD, pre: str, mel_norm: Tensor, cond, periods=(1, 2, 3, 5), buffers=None, training=True):
    """MultiPeriodMelDiscriminator.forward (lab3_models.py:353-375)."""
    logits, feats = [], []
    for i, p in enumerate(periods):
        x = mel_norm[:, :, ::max(1, int(p))]
        if x.shape[-1] < 16:
            x = mel_norm
        li, fi = mel_disc_forward(sd, f"{pre}discriminators.{i}.", x, cond, buffers, training)
        logits.append(li)
        feats.append(fi)
    return torch.stack(logits, dim=0).mean(dim=0), torch.cat(feats, dim=-1)


def hybrid_forward(sd: SD, mel_norm: Tensor, cond, num_scales: int = 3, periods=(1, 2, 3, 5), buffers=None, training=True):
    """HybridMelDiscriminator.forward (lab3_models.py:394-408)."""
    l_ms, f_ms = multiscale_forward(sd, "ms.", mel_norm, cond, num_scales, buffers, training)
    l_mp, f_mp = multiperiod_forward(sd, "mp.", mel_norm, cond, periods, buffers, training)
    return 0.5 * (l_ms + l_mp), torch.cat([f_ms, f_mp], dim=-1)


# ------------------------------------------------------------------------------------------------ losses
def bce_logits(logits: Tensor, target_is_real: bool, real_label: float = 1.0, fake_label: float = 0.0) -> Tensor:
    """lab3_models.py:430-438."""
    t = torch.full_like(logits, float(real_label) if target_is_real else float(fake_label))
    return F.binary_cross_entropy_with_logits(logits, t)


def gan_d_loss(gan_loss: str, logits_real: Tensor, logits_fake: Tensor, d_real_label: float = 1.0, d_fake_label: float = 0.0) -> Tensor:
    """lab3_train.py:297-312."""
    key = str(gan_loss).strip().lower()
    if key == "hinge":
        return 0.5 * (F.relu(1.0 - logits_real).mean() + F.relu(1.0 + logits_fake).mean())
    if key == "bce":
        return 0.5 * (bce_logits(logits_real, True, d_real_label, d_fake_label) + bce_logits(logits_fake, False, d_real_label, d_fake_label))
    raise ValueError(f"Unsupported gan_loss: {gan_loss}")


def gan_g_loss(gan_loss: str, logits_fake: Tensor, g_real_label: float = 1.0, d_fake_label: float = 0.0) -> Tensor:
    """lab3_train.py:315-326."""
    key = str(gan_loss).strip().lower()
    if key == "hinge":
        return -logits_fake.mean()
    if key == "bce":
        return bce_logits(logits_fake, True, g_real_label, d_fake_label)
    raise ValueError(f"Unsupported gan_loss: {gan_loss}")


def denormalize_log_mel(mel_norm: Tensor) -> Tensor:
    """lab3_bridge.py:135-139."""
    x = (torch.clamp(mel_norm, -1.0, 1.0) + 1.0) * 0.5
    return x * (MEL_DB_MAX - MEL_DB_MIN) + MEL_DB_MIN


# ------------------------------------------------------------------------------------------------ one stage-1 iteration
@dataclass
class MelWeights:  # the TrainWeights fields train_stage1 reads (lab3_train.py:32-37, 437-445)
    adv: float = 0.5
    recon_l1: float = 8.0
    content: float = 2.0
    style: float = 1.0
    continuity: float = 1.0
    mrstft: float = 0.0


@dataclass
class MelStepCfg:  # the StageTrainConfig fields train_stage1 reads (lab3_train.py:59-126)
    gan_loss: str = "bce"
    disc_use_content_cond: bool = False
    weights: MelWeights = field(default_factory=MelWeights)
    d_real_label: float = 1.0
    d_fake_label: float = 0.0
    g_real_label: float = 1.0
    d_step_period: int = 1
    g_grad_clip_norm: float = 0.0
    d_grad_clip_norm: float = 0.0
    lr_g: float = 2e-4
    lr_d: float = 2e-4
    betas: Tuple[float, float] = (0.5, 0.999)
    mrstft_resolutions: Tuple[Tuple[int, int, int], ...] = ((64, 16, 64), (128, 32, 128), (256, 64, 256))
    decoder: DecoderCfg = field(default_factory=DecoderCfg)


def adam_state(lr: float, betas) -> AdamWState:
    """torch.optim.Adam(lr, betas) of train_stage1 (lab3_train.py:357-358): AdamW arithmetic with weight decay 0."""
    st = AdamWState()
    st.lr, st.betas, st.weight_decay = float(lr), (float(betas[0]), float(betas[1])), 0.0
    return st


def _leaf(sd: SD) -> SD:
    return {k: (v.detach().clone().requires_grad_(True) if torch.is_floating_point(v) and "running_" not in k else v) for k, v in sd.items()}


def stage1_iteration(g_sd: SD, d_sd: SD, opt_g: AdamWState, opt_d: AdamWState, real: Tensor, zc: Tensor, cond: Tensor, step_idx: int,
                     cfg: MelStepCfg = MelStepCfg(), record: bool = False) -> Dict[str, object]:
    """One pass of train_stage1's loop body (lab3_train.py:380-456) with a plain MelDiscriminator, updating g_sd / d_sd (and D's
    BatchNorm buffers) in place.  The content / style terms need the caller's frozen Lab1 encoder (lab3_train.py:432-435) and are
    left out, as the codec oracle leaves them out: weights.content / .style multiply zero.  R1 (lab3_train.py:402-404, a double
    backward through BatchNorm) is not restated."""
    out: Dict[str, object] = {}
    cond_d = torch.cat([cond, zc], dim=1) if cfg.disc_use_content_cond else cond
    do_d_step = (step_idx % max(1, int(cfg.d_step_period))) == 0
    gl, dl = _leaf(g_sd), _leaf(d_sd)
    bufs: SD = {}
    # D step (:391-415).  All three D forwards of the iteration run in training mode and each updates the running statistics.
    with torch.no_grad():
        fake_det = decoder_forward(gl, zc, cond, cfg.decoder)
    logit_real, _ = mel_disc_forward(dl, "", real, cond_d, bufs)
    for k, v in bufs.items():
        dl[k] = v
    logit_fake, _ = mel_disc_forward(dl, "", fake_det, cond_d, bufs)
    for k, v in bufs.items():
        dl[k] = v
    loss_d = gan_d_loss(cfg.gan_loss, logit_real, logit_fake, cfg.d_real_label, cfg.d_fake_label)
    out["loss_d"] = float(loss_d)
    if do_d_step:
        names = [k for k, v in dl.items() if torch.is_floating_point(v) and v.requires_grad]
        grads = dict(zip(names, torch.autograd.grad(loss_d, [dl[k] for k in names], allow_unused=True)))
        grads = {k: (g if g is not None else torch.zeros_like(dl[k])) for k, g in grads.items()}
        if cfg.d_grad_clip_norm > 0.0:
            out["d_grad_norm"] = clip_grad_norm(grads, cfg.d_grad_clip_norm)
        if record:
            out["d_grads"] = {k: g.clone() for k, g in grads.items()}
        with torch.no_grad():
            params = {k: dl[k].detach() for k in names}
            adamw_step(params, grads, opt_d)
            for k in names:
                dl[k] = params[k].clone().requires_grad_(True)
    # G step (:418-451)
    fake = decoder_forward(gl, zc, cond, cfg.decoder)
    dfrozen = {k: (v.detach() if torch.is_floating_point(v) else v) for k, v in dl.items()}
    logit_fake_g, _ = mel_disc_forward(dfrozen, "", fake, cond_d, bufs)
    for k, v in bufs.items():
        dl[k] = v
    loss_adv = gan_g_loss(cfg.gan_loss, logit_fake_g, cfg.g_real_label, cfg.d_fake_label)
    loss_l1 = F.l1_loss(fake, real)
    loss_cont = mel_mrstft_loss(fake, real, cfg.mrstft_resolutions)
    w = cfg.weights
    loss_g = w.adv * loss_adv + w.recon_l1 * loss_l1 + w.continuity * loss_cont + w.mrstft * loss_cont
    gnames = [k for k, v in gl.items() if torch.is_floating_point(v) and v.requires_grad]
    ggrads = dict(zip(gnames, torch.autograd.grad(loss_g, [gl[k] for k in gnames], allow_unused=True)))
    ggrads = {k: (g if g is not None else torch.zeros_like(gl[k])) for k, g in ggrads.items()}
    if cfg.g_grad_clip_norm > 0.0:
        out["g_grad_norm"] = clip_grad_norm(ggrads, cfg.g_grad_clip_norm)
    if record:
        out["g_grads"] = {k: g.clone() for k, g in ggrads.items()}
        out["fake"] = fake.detach().clone()
    with torch.no_grad():
        params = {k: gl[k].detach() for k in gnames}
        adamw_step(params, ggrads, opt_g)
        for k in gnames:
            g_sd[k] = params[k].clone()
        for k, v in dl.items():
            d_sd[k] = v.detach().clone() if torch.is_tensor(v) else v
    out.update(loss_g=float(loss_g), loss_adv=float(loss_adv), loss_l1=float(loss_l1), loss_continuity=float(loss_cont),
               loss_mrstft=float(loss_cont))
    return out
