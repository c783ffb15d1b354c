"""Mel-track stage-1 training of the reference (lab 3/src/lab3_train.py) on the HIP-backed models of gan_amd/lab3_models.py.

SURVEY.md section 8(f) next-4, first slice: `TrainWeights` (:31-56), `StageTrainConfig` (:58-126), `_gan_d_loss` / `_gan_g_loss`
(:297-326) and `train_stage1` (:342-483) with the reference's names, arguments and history rows; one pass of its loop body is
`mel_stage1_iteration`.  Adam (torch.optim.Adam at :357-358) and the optional gradient clipping (:409-410, :447-448) run as the fused
clip + AdamW kernel with weight decay 0.  Not in this slice: `train_stage2` (:486-1088, the 23-term stage) and R1 (:329-339,
:402-404: a double backward through BatchNorm) -- asking for either raises.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .lab3_losses import multi_resolution_stft_loss
from .lab3_models import bce_logits
from .optim import FusedAdamW

MEL_DB_MIN, MEL_DB_MAX = -80.0, 0.0  # lab3_bridge.py (denormalize_log_mel, :135-139)


@dataclass
class TrainWeights:
    """reference lab3_train.py:31-56 (stage 1 reads adv, recon_l1, content, style, continuity, mrstft; the rest belong to stage 2)."""
    adv: float = 0.5
    recon_l1: float = 8.0
    content: float = 2.0
    style: float = 1.0
    continuity: float = 1.0
    mrstft: float = 0.0
    flatness: float = 0.5
    feature_match: float = 0.0
    perceptual: float = 0.0
    style_hinge: float = 0.0
    contrastive: float = 0.0
    batch_infonce_diversity: float = 0.0
    pfm_style: float = 0.0
    diversity: float = 0.0
    timbre_balance: float = 0.0
    lowmid_recon: float = 0.0
    spectral_tilt: float = 0.0
    zcr_proxy: float = 0.0
    hf_muzzle: float = 0.0
    highpass_anchor: float = 0.0
    style_mid: float = 0.0
    mel_diversity: float = 0.0
    target_profile: float = 0.0


@dataclass
class StageTrainConfig:
    """The fields of the reference's StageTrainConfig (lab3_train.py:58-126) that stage 1 reads, with its defaults; stage-2-only
    controls are accepted through `extra` so that a reference call site's keyword arguments do not fail."""
    stage_name: str
    epochs: int
    lr_g: float = 2e-4
    lr_d: float = 2e-4
    betas: Tuple[float, float] = (0.5, 0.999)
    max_batches_per_epoch: Optional[int] = None
    gan_loss: str = "bce"
    disc_use_content_cond: bool = False
    weights: TrainWeights = field(default_factory=TrainWeights)
    d_real_label: float = 1.0
    d_fake_label: float = 0.0
    g_real_label: float = 1.0
    r1_gamma: float = 0.0
    r1_interval: int = 16
    d_step_period: int = 1
    g_grad_clip_norm: float = 0.0
    d_grad_clip_norm: float = 0.0
    mrstft_resolutions: Tuple[Tuple[int, int, int], ...] = ((64, 16, 64), (128, 32, 128), (256, 64, 256))
    extra: Dict[str, object] = field(default_factory=dict)


def _gan_d_loss(gan_loss: str, logits_real: torch.Tensor, logits_fake: torch.Tensor, d_real_label: float, d_fake_label: float) -> torch.Tensor:
    """reference lab3_train.py:297-312."""
    key = str(gan_loss).strip().lower()
    if key == "hinge":
        return 0.5 * (F.relu(1.0 - logits_real).mean() + F.relu(1.0 + logits_fake).mean())
    if key == "bce":
        return 0.5 * (bce_logits(logits_real, True, real_label=float(d_real_label), fake_label=float(d_fake_label))
                      + bce_logits(logits_fake, False, real_label=float(d_real_label), fake_label=float(d_fake_label)))
    raise ValueError(f"Unsupported gan_loss: {gan_loss}")


def _gan_g_loss(gan_loss: str, logits_fake: torch.Tensor, g_real_label: float, d_fake_label: float) -> torch.Tensor:
    """reference lab3_train.py:315-326."""
    key = str(gan_loss).strip().lower()
    if key == "hinge":
        return -logits_fake.mean()
    if key == "bce":
        return bce_logits(logits_fake, True, real_label=float(g_real_label), fake_label=float(d_fake_label))
    raise ValueError(f"Unsupported gan_loss: {gan_loss}")


def denormalize_log_mel(mel_norm: torch.Tensor) -> torch.Tensor:
    """reference lab3_bridge.py:135-139: [-1, 1] -> dB."""
    return (torch.clamp(mel_norm, -1.0, 1.0) + 1.0) * (0.5 * (MEL_DB_MAX - MEL_DB_MIN)) + MEL_DB_MIN


def _cosine_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """reference lab3_losses.py (cosine_distance): 1 - cos."""
    return 1.0 - F.cosine_similarity(a, b, dim=-1)


def _backward_step(loss: torch.Tensor, opt: FusedAdamW, clip: float) -> None:
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.max_grad_norm = float(clip) if float(clip) > 0.0 else 0.0
    opt.step()


def mel_stage1_iteration(stage_cfg: StageTrainConfig, generator, discriminator, opt_g: FusedAdamW, opt_d: FusedAdamW,
                         real: torch.Tensor, zc: torch.Tensor, cond: torch.Tensor, step_idx: int, zs: Optional[torch.Tensor] = None,
                         frozen_encoder=None) -> Dict[str, torch.Tensor]:
    """One pass of train_stage1's loop body (reference lab3_train.py:380-456).  Returns device scalars (no host sync).
    frozen_encoder (+ zs): the caller's frozen Lab1 encoder for the content / style terms (:432-435); without it they stay at
    zero and nonzero weights for them are an error, not a silent drop."""
    if float(stage_cfg.r1_gamma) > 0.0:
        raise NotImplementedError("R1 on the mel discriminator (a double backward through BatchNorm) is not implemented")
    w = stage_cfg.weights
    if frozen_encoder is None and (float(w.content) != 0.0 or float(w.style) != 0.0) and not stage_cfg.extra.get("allow_missing_encoder", False):
        raise ValueError("weights.content / weights.style need the frozen Lab1 encoder (pass frozen_encoder, or set them to 0)")
    cond_d = torch.cat([cond, zc], dim=1) if bool(stage_cfg.disc_use_content_cond) else cond
    do_d_step = (int(step_idx) % int(max(1, stage_cfg.d_step_period))) == 0
    out: Dict[str, torch.Tensor] = {}
    # D step (:391-415)
    with torch.no_grad():
        fake_det = generator(zc, cond)
    loss_d = _gan_d_loss(stage_cfg.gan_loss, discriminator(real, cond_d), discriminator(fake_det, cond_d),
                         float(stage_cfg.d_real_label), float(stage_cfg.d_fake_label))
    if do_d_step:
        _backward_step(loss_d, opt_d, stage_cfg.d_grad_clip_norm)
    # G step (:418-451)
    fake = generator(zc, cond)
    d_params = list(discriminator.parameters())
    saved = [p.requires_grad for p in d_params]
    for p in d_params:  # D's own gradients of this pass are never used (zero_grad(set_to_none) precedes its next backward)
        p.requires_grad_(False)
    try:
        logit_fake_g = discriminator(fake, cond_d)
    finally:
        for p, r in zip(d_params, saved):
            p.requires_grad_(r)
    loss_adv = _gan_g_loss(stage_cfg.gan_loss, logit_fake_g, float(stage_cfg.g_real_label), float(stage_cfg.d_fake_label))
    loss_l1 = F.l1_loss(fake, real)
    loss_continuity = multi_resolution_stft_loss(fake, real, resolutions=stage_cfg.mrstft_resolutions)
    loss_mrstft = loss_continuity
    zero = torch.zeros((), device=real.device)
    loss_content = loss_style = zero
    if frozen_encoder is not None:
        enc = frozen_encoder.forward_log_mel_tensor(denormalize_log_mel(fake))
        loss_content = _cosine_distance(enc["z_content"], zc).mean()
        if zs is not None:
            loss_style = _cosine_distance(enc["z_style"], zs).mean()
    loss_g = (w.adv * loss_adv + w.recon_l1 * loss_l1 + w.content * loss_content + w.style * loss_style
              + w.continuity * loss_continuity + w.mrstft * loss_mrstft)
    _backward_step(loss_g, opt_g, stage_cfg.g_grad_clip_norm)
    out.update(loss_g=loss_g.detach(), loss_d=loss_d.detach(), loss_r1=zero, loss_adv=loss_adv.detach(), loss_l1=loss_l1.detach(),
               loss_content=loss_content.detach(), loss_style=loss_style.detach(), loss_continuity=loss_continuity.detach(),
               loss_mrstft=loss_mrstft.detach(), d_step=torch.tensor(1.0 if do_d_step else 0.0))
    return out


def _save_stage_checkpoint(path: Path, epoch: int, generator, discriminator, opt_g, opt_d, meta: Optional[Dict] = None) -> None:
    """reference lab3_train.py:207-237: the same dictionary keys (optimizer states in torch.optim.Adam's layout)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save({"epoch": int(epoch), "generator": generator.state_dict(), "discriminator": discriminator.state_dict(),
                "opt_g": opt_g.state_dict(), "opt_d": opt_d.state_dict(), "meta": meta or {}}, path)


def _load_stage_checkpoint(path: Path, generator, discriminator, opt_g, opt_d) -> int:
    """reference lab3_train.py:240-276: epoch to resume from (0 when there is no checkpoint)."""
    path = Path(path)
    if not path.exists():
        return 0
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    generator.load_state_dict(ckpt["generator"], strict=True)
    discriminator.load_state_dict(ckpt["discriminator"], strict=True)
    if "opt_g" in ckpt:
        opt_g.load_state_dict(ckpt["opt_g"])
    if "opt_d" in ckpt:
        opt_d.load_state_dict(ckpt["opt_d"])
    return int(ckpt.get("epoch", 0))


def train_stage1(generator, discriminator, frozen_encoder, train_loader, cond_bank: torch.Tensor, device, stage_cfg: StageTrainConfig,
                 checkpoint_path: Path, resume: bool = False) -> List[Dict]:
    """reference lab3_train.py:342-483: same arguments, same per-epoch history rows (nine loss means, stage, epoch, d-step counts)."""
    generator.train()
    discriminator.train()
    cond_bank = cond_bank.to(device)
    opt_g = FusedAdamW(generator.parameters(), lr=stage_cfg.lr_g, betas=stage_cfg.betas, weight_decay=0.0)
    opt_d = FusedAdamW(discriminator.parameters(), lr=stage_cfg.lr_d, betas=stage_cfg.betas, weight_decay=0.0)
    start_epoch = _load_stage_checkpoint(checkpoint_path, generator, discriminator, opt_g, opt_d) if resume else 0
    keys = ("loss_g", "loss_d", "loss_r1", "loss_adv", "loss_l1", "loss_content", "loss_style", "loss_continuity", "loss_mrstft")
    history: List[Dict] = []
    step_idx = 0
    for epoch in range(start_epoch, int(stage_cfg.epochs)):
        sums = torch.zeros(len(keys), device=device)
        nb = taken = 0
        for bidx, batch in enumerate(train_loader):
            if stage_cfg.max_batches_per_epoch is not None and bidx >= int(stage_cfg.max_batches_per_epoch):
                break
            real = batch["mel_norm"].to(device).float()
            zc = batch["z_content"].to(device).float()
            zs = batch["z_style"].to(device).float()
            cond = cond_bank[batch["genre_idx"].to(device).long()]
            out = mel_stage1_iteration(stage_cfg, generator, discriminator, opt_g, opt_d, real, zc, cond, step_idx, zs=zs,
                                       frozen_encoder=frozen_encoder)
            sums += torch.stack([out[k].float().reshape(()) for k in keys])  # one host read per epoch, not nine per iteration
            taken += int((step_idx % int(max(1, stage_cfg.d_step_period))) == 0)
            nb += 1
            step_idx += 1
        means = (sums / max(1, nb)).tolist()
        ep: Dict = {k: float(v) for k, v in zip(keys, means)}
        ep.update(stage="stage1", epoch=int(epoch + 1), d_steps_taken=float(taken), d_steps_skipped=float(nb - taken),
                  d_step_rate=float(taken / max(1, nb)))
        history.append(ep)
        _save_stage_checkpoint(checkpoint_path, int(epoch + 1), generator, discriminator, opt_g, opt_d, meta={"stage": "stage1"})
        print(f"[stage1] epoch={epoch + 1} loss_g={ep['loss_g']:.4f} loss_d={ep['loss_d']:.4f}")
    return history
