"""Mel-track training of the reference (lab 3/src/lab3_train.py) on the HIP-backed models of gan_amd/lab3_models.py.

SURVEY.md section 8(f) next-4: `TrainWeights` (:31-56), `StageTrainConfig` (:58-126), the bank builders and target samplers
(:129-204), `_gan_d_loss` / `_gan_g_loss` (:297-326), `train_stage1` (:342-483) and `train_stage2` (:486-1088) with the reference's
names, arguments, history rows and checkpoint keys; one pass of a loop body is `mel_stage1_iteration` / `Stage2Runner.iteration`.
Adam (torch.optim.Adam at :357-358, :538-540) and the optional gradient clipping run as the fused clip + AdamW kernel with weight
decay 0; the optimizer of the caller's thawed encoder parameters (:532-536) stays torch.optim.Adam, on the caller's own module.
R1 (:329-339) differentiates the discriminator's input gradient: inside `_r1_penalty` the 2-D convolutions run as twice-differentiable
nodes (ops.ConvTwice: the same three HIP kernels in permuted roles), BatchNorm's double backward is torch's.
"""
from __future__ import annotations

import contextlib

import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import lab3_losses as LL
from .conv2d import twice_differentiable
from .lab3_losses import multi_resolution_stft_loss
from .lab3_models import StyleCritic, bce_logits
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
    """reference lab3_train.py:58-126, field for field with its defaults (`extra` is this build's: test switches only)."""
    stage_name: str
    epochs: int
    lr_g: float = 2e-4
    lr_d: float = 2e-4
    betas: Tuple[float, float] = (0.5, 0.999)
    max_batches_per_epoch: Optional[int] = None
    gan_loss: str = "bce"
    disc_use_content_cond: bool = False
    weights: TrainWeights = field(default_factory=TrainWeights)
    # stage-2 controls (stage 1 ignores them)
    content_weight_start: Optional[float] = None
    content_weight_end: Optional[float] = None
    style_label_smoothing: float = 0.0
    style_only_warmup_epochs: int = 0
    g_lr_warmup_epochs: int = 0
    g_lr_start_mult: float = 1.0
    cond_noise_std: float = 0.0
    style_jitter_std: float = 0.0
    cond_mode: str = "mix"  # centroid | exemplar | mix
    cond_alpha_start: float = 0.8
    cond_alpha_end: float = 0.4
    cond_exemplar_noise_std: float = 0.03
    target_balance: bool = True
    style_hinge_target_conf: float = 0.85
    adaptive_content_weight: bool = False
    adaptive_content_low: float = 0.05
    adaptive_content_high: float = 0.5
    adaptive_conf_low: float = 0.5
    adaptive_conf_high: float = 0.8
    style_class_balance: bool = False
    style_critic_lr: float = 2e-4
    contrastive_temp: float = 0.10
    diversity_margin: float = 0.90
    diversity_max_pairs: int = 64
    style_lowpass_keep_bins: int = 80
    style_lowpass_cutoff_hz: Optional[float] = None
    style_mid_low_bin: int = 8
    style_mid_high_bin: int = 56
    lowmid_split_bin: int = 80
    lowmid_gain: float = 5.0
    high_gain: float = 0.5
    spectral_tilt_max_ratio: float = 0.7
    zcr_proxy_target_max: float = 0.18
    hf_top_frac: float = 0.20
    hf_max_ratio_multiplier: float = 1.20
    highpass_cut_bin: int = 12
    highpass_gain: float = 2.0
    mel_diversity_margin: float = 0.85
    mel_diversity_max_pairs: int = 64
    d_real_label: float = 1.0
    d_fake_label: float = 0.0
    g_real_label: float = 1.0
    r1_gamma: float = 0.0
    r1_interval: int = 16
    d_loss_floor_for_step: float = 0.0
    d_step_period: int = 1
    g_grad_clip_norm: float = 0.0
    d_grad_clip_norm: float = 0.0
    style_thaw_last_epochs: int = 0
    style_thaw_lr: float = 1e-6
    style_thaw_scope: str = "style_head"
    batch_infonce_temp: float = 0.15
    mrstft_resolutions: Tuple[Tuple[int, int, int], ...] = ((64, 16, 64), (128, 32, 128), (256, 64, 256))
    extra: Dict[str, object] = field(default_factory=dict)


def load_target_centroids(target_centroids_json: Path) -> Dict[str, np.ndarray]:
    """reference lab3_train.py:129-146: {genre: unit vector} from a JSON of {"vector" | "vector160": [...]} records."""
    path = Path(target_centroids_json)
    if not path.exists():
        raise FileNotFoundError(f"Target centroid file not found: {path}")
    records = json.loads(path.read_text(encoding="utf-8"))
    out: Dict[str, np.ndarray] = {}
    for genre, rec in records.items():
        key = "vector" if "vector" in rec else "vector160" if "vector160" in rec else None
        if key is None:
            raise ValueError(f"Missing vector field for genre {genre}")
        v = np.asarray(rec[key], dtype=np.float32)
        out[str(genre)] = v / (np.linalg.norm(v) + 1e-8)
    return out


def build_condition_bank(genre_to_idx: Dict[str, int], centroid_map: Dict[str, np.ndarray]) -> torch.Tensor:
    """reference :149-157: [n_genres, dim], row idx = that genre's centroid."""
    missing = [g for g in genre_to_idx if g not in centroid_map]
    if missing:
        raise ValueError(f"Centroid map missing genres: {missing}")
    bank = np.zeros((len(genre_to_idx), len(next(iter(centroid_map.values())))), dtype=np.float32)
    for genre, idx in genre_to_idx.items():
        bank[int(idx)] = centroid_map[genre]
    return torch.from_numpy(bank)


def build_style_centroid_bank(z_style: np.ndarray, genre_idx: np.ndarray, n_genres: int) -> torch.Tensor:
    """reference :160-168: per-genre mean of z_style, L2-normalised; zero rows for absent genres."""
    bank = np.zeros((int(n_genres), z_style.shape[1]), dtype=np.float32)
    for g in range(int(n_genres)):
        rows = z_style[genre_idx == g]
        if len(rows):
            v = rows.mean(axis=0).astype(np.float32)
            bank[g] = v / (np.linalg.norm(v) + 1e-8)
    return torch.from_numpy(bank)


def build_style_exemplar_bank(z_style: np.ndarray, genre_idx: np.ndarray, n_genres: int) -> Dict[int, torch.Tensor]:
    """reference :171-183: {genre: its L2-normalised z_style rows} for the genres that occur."""
    bank: Dict[int, torch.Tensor] = {}
    for g in range(int(n_genres)):
        rows = z_style[genre_idx == g].astype(np.float32)
        if len(rows):
            bank[g] = torch.from_numpy(rows / (np.linalg.norm(rows, axis=1, keepdims=True) + 1e-8))
    return bank


def _bump_clashes(tgt: torch.Tensor, source_idx: torch.Tensor, n_genres: int) -> torch.Tensor:
    return torch.where(tgt == source_idx, (tgt + 1) % int(n_genres), tgt)


def _sample_shift_targets(source_idx: torch.Tensor, n_genres: int) -> torch.Tensor:
    """reference :186-193: uniform targets, moved one genre on where they hit the source."""
    tgt = torch.randint(low=0, high=int(n_genres), size=(int(source_idx.shape[0]),), device=source_idx.device)
    return _bump_clashes(tgt, source_idx, n_genres)


def _sample_shift_targets_balanced(source_idx: torch.Tensor, n_genres: int, offset: int = 0) -> torch.Tensor:
    """reference :196-204: round-robin targets starting at `offset`, same clash rule."""
    tgt = (torch.arange(int(source_idx.shape[0]), device=source_idx.device, dtype=torch.long) + int(offset)) % int(n_genres)
    return _bump_clashes(tgt, source_idx, n_genres)


def _linear_weight(start: float, end: float, epoch_idx: int, total_epochs: int) -> float:
    """reference :285-289."""
    if total_epochs <= 1:
        return float(end)
    return float(start + (end - start) * (float(epoch_idx) / float(total_epochs - 1)))


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


def _r1_penalty(discriminator, real: torch.Tensor, cond: Optional[torch.Tensor]) -> torch.Tensor:
    """reference lab3_train.py:329-339: mean over the batch of || d sum(logits) / d real ||^2, with a graph behind it."""
    real_req = real.detach().requires_grad_(True)
    with twice_differentiable():
        logits = discriminator(real_req, cond)
        grad, = torch.autograd.grad(outputs=logits.sum(), inputs=real_req, create_graph=True, retain_graph=True, only_inputs=True)
    return grad.pow(2).reshape(grad.shape[0], -1).sum(dim=1).mean()


def denormalize_log_mel(mel_norm: torch.Tensor) -> torch.Tensor:
    """reference lab3_bridge.py:135-139: [-1, 1] -> dB."""
    return (torch.clamp(mel_norm, -1.0, 1.0) + 1.0) * (0.5 * (MEL_DB_MAX - MEL_DB_MIN)) + MEL_DB_MIN


def _cosine_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """reference lab3_losses.py (cosine_distance): 1 - cos."""
    return 1.0 - F.cosine_similarity(a, b, dim=-1)


def _backward_step(loss: torch.Tensor, opt: FusedAdamW, clip: float, dp=None) -> None:
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.max_grad_norm = float(clip) if float(clip) > 0.0 else 0.0
    if dp is None:
        opt.step()
        return
    opt.gather_grads()
    dp.allreduce_flat(opt)  # SUM over ranks in buckets; the mean and the common clip coefficient are applied by the AdamW launch
    opt.step(grads_gathered=True)


def mel_stage1_iteration(stage_cfg: StageTrainConfig, generator, discriminator, opt_g: FusedAdamW, opt_d: FusedAdamW,
                         real: torch.Tensor, zc: torch.Tensor, cond: torch.Tensor, step_idx: int, zs: Optional[torch.Tensor] = None,
                         frozen_encoder=None, dp=None) -> Dict[str, torch.Tensor]:
    """One pass of train_stage1's loop body (reference lab3_train.py:380-456).  Returns device scalars (no host sync).
    frozen_encoder (+ zs): the caller's frozen Lab1 encoder for the content / style terms (:432-435); without it they stay at
    zero and nonzero weights for them are an error, not a silent drop.  dp (gan_amd.dp.DataParallel, not in the reference): this
    rank's shard of the batch; gradients are all-reduced before each optimizer step, the MR-STFT ratio is the global batch's, and the
    discriminator's BatchNorm statistics are this rank's own (what DistributedDataParallel without SyncBN conversion does)."""
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
    zero = torch.zeros((), device=real.device)
    r1 = zero
    if do_d_step:
        if float(stage_cfg.r1_gamma) > 0.0 and (int(step_idx) % int(max(1, stage_cfg.r1_interval)) == 0):
            r1 = 0.5 * float(stage_cfg.r1_gamma) * _r1_penalty(discriminator, real, cond_d)
            loss_d = loss_d + r1
        _backward_step(loss_d, opt_d, stage_cfg.d_grad_clip_norm, dp)
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
    loss_continuity = multi_resolution_stft_loss(fake, real, resolutions=stage_cfg.mrstft_resolutions, precision=getattr(generator, "precision", "fp32"),
                                                 sum_hook=None if dp is None else dp.stft_sum_hook)
    loss_mrstft = loss_continuity
    loss_content = loss_style = zero
    if frozen_encoder is not None:
        enc = frozen_encoder.forward_log_mel_tensor(denormalize_log_mel(fake))
        loss_content = _cosine_distance(enc["z_content"], zc).mean()
        if zs is not None:
            loss_style = _cosine_distance(enc["z_style"], zs).mean()
    loss_g = (w.adv * loss_adv + w.recon_l1 * loss_l1 + w.content * loss_content + w.style * loss_style
              + w.continuity * loss_continuity + w.mrstft * loss_mrstft)
    _backward_step(loss_g, opt_g, stage_cfg.g_grad_clip_norm, dp)
    out.update(loss_g=loss_g.detach(), loss_d=loss_d.detach(), loss_r1=r1.detach(), loss_adv=loss_adv.detach(), loss_l1=loss_l1.detach(),
               loss_content=loss_content.detach(), loss_style=loss_style.detach(), loss_continuity=loss_continuity.detach(),
               loss_mrstft=loss_mrstft.detach(), d_step=torch.tensor(1.0 if do_d_step else 0.0))
    return out


class GraphedStage1:
    """`mel_stage1_iteration` recorded as ONE HIP graph per step variant and replayed (torch.cuda.CUDAGraph; no tracing compiler: the
    graph is the iteration's own launches).  Why: at the reference's default batch of 32 the iteration is ~650 launches of a few
    microseconds each and the host cannot issue them as fast as the GPU runs them (10.5 ms eager against 6.8 ms replayed; batch 16:
    10.9 against 5.4; from batch 256 up there is nothing to gain).  What makes a replay equal to an eager iteration: the batch goes
    through fixed input buffers; AdamW's step number is read from device memory (`FusedAdamW.use_device_step`); the packed-weight cache
    is emptied before recording, so the recording contains every pack it reads; BatchNorm's counters are device ops already.  Eager
    fallback, same results: the first `warmup` calls of a variant (lazy initialisation happens there), a batch of another size (the
    last one of an epoch), R1 steps (double backward), data parallelism (collectives), and a frozen encoder that is not this package's.
    `train_stage1` uses it unless `stage_cfg.extra["hip_graph"]` is False; or call it like `mel_stage1_iteration`."""

    def __init__(self, stage_cfg: StageTrainConfig, generator, discriminator, opt_g: FusedAdamW, opt_d: FusedAdamW, frozen_encoder=None,
                 warmup: int = 2):
        self.cfg, self.G, self.D, self.og, self.od, self.enc = stage_cfg, generator, discriminator, opt_g, opt_d, frozen_encoder
        self.warmup = int(warmup)
        self.max_graphs = 4
        self.graphs: Dict = {}   # (do_d_step, batch shape) -> (graph, static inputs, static outputs)
        self.seen: Dict = {}
        from .lab3_bridge import FrozenLab1Encoder
        self.capturable = frozen_encoder is None or isinstance(frozen_encoder, FrozenLab1Encoder)
        opt_g.use_device_step()
        opt_d.use_device_step()

    def _eager(self, real, zc, cond, step_idx, zs):
        return mel_stage1_iteration(self.cfg, self.G, self.D, self.og, self.od, real, zc, cond, step_idx, zs=zs, frozen_encoder=self.enc)

    def __call__(self, real: torch.Tensor, zc: torch.Tensor, cond: torch.Tensor, step_idx: int, zs: Optional[torch.Tensor] = None,
                 dp=None) -> Dict[str, torch.Tensor]:
        cfg = self.cfg
        do_d = (int(step_idx) % int(max(1, cfg.d_step_period))) == 0
        do_r1 = do_d and float(cfg.r1_gamma) > 0.0 and (int(step_idx) % int(max(1, cfg.r1_interval)) == 0)
        if dp is not None:
            return mel_stage1_iteration(cfg, self.G, self.D, self.og, self.od, real, zc, cond, step_idx, zs=zs, frozen_encoder=self.enc, dp=dp)
        if do_r1 or not self.capturable or not real.is_cuda:
            return self._eager(real, zc, cond, step_idx, zs)
        key = (do_d, tuple(real.shape), tuple(zc.shape), tuple(cond.shape), None if zs is None else tuple(zs.shape))
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if n < self.warmup or (key not in self.graphs and len(self.graphs) >= self.max_graphs):
            return self._eager(real, zc, cond, step_idx, zs)  # (every recording keeps its activations: a loader of ever-changing shapes stays eager)
        from . import ops
        if key not in self.graphs:
            ins = [t.clone() if t is not None else None for t in (real, zc, cond, zs)]
            ops._PACK_CACHE.clear()  # the recording must contain every weight pack it reads ...
            ops._ARENA.clear()       # ... and the fill of every pre-zeroed chunk it carves accumulators from (a chunk left over from an
            ops._CIN1_CACHE.clear()  # earlier recording is zeroed by THAT graph's fill node only), and every padded input copy
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            steps_before = (self.og.step_count, self.od.step_count)
            with torch.cuda.graph(graph):
                outs = mel_stage1_iteration(cfg, self.G, self.D, self.og, self.od, ins[0], ins[1], ins[2], int(step_idx), zs=ins[3],
                                            frozen_encoder=self.enc)
            # recording launches nothing: undo the host-side step counts its optimizer calls made (the device counters did not move)
            self.og.step_count, self.od.step_count = steps_before
            ops._ARENA.clear()       # nobody else carves from this graph's chunk
            ops._CIN1_CACHE.clear()
            self.graphs[key] = (graph, ins, outs)
        graph, ins, outs = self.graphs[key]
        for dst, src in zip(ins, (real, zc, cond, zs)):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        graph.replay()
        self.og.step_count += 1
        self.og._epoch[0] += 1
        if do_d:
            self.od.step_count += 1
            self.od._epoch[0] += 1
        ops._PACK_CACHE.clear()  # packs made inside the graph's memory are not valid cache entries for eager calls
        return outs


def _save_stage_checkpoint(path: Path, epoch: int, generator, discriminator, opt_g, opt_d, meta: Optional[Dict] = None, style_critic=None,
                           opt_sc=None, frozen_encoder_model=None, opt_style_thaw=None) -> None:
    """reference lab3_train.py:207-237: the same dictionary keys (optimizer states in torch.optim.Adam's layout)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    payload = {"epoch": int(epoch), "generator": generator.state_dict(), "discriminator": discriminator.state_dict(),
               "opt_g": opt_g.state_dict(), "opt_d": opt_d.state_dict(), "meta": meta or {}}
    for key, obj in (("style_critic", style_critic), ("opt_sc", opt_sc), ("frozen_encoder_model", frozen_encoder_model),
                     ("opt_style_thaw", opt_style_thaw)):
        if obj is not None:
            payload[key] = obj.state_dict()
    torch.save(payload, path)


def _load_stage_checkpoint(path: Path, generator, discriminator, opt_g, opt_d, style_critic=None, opt_sc=None, frozen_encoder_model=None,
                           opt_style_thaw=None) -> int:
    """reference lab3_train.py:240-276: epoch to resume from (0 when there is no checkpoint).  Read with weights_only=True."""
    path = Path(path)
    if not path.exists():
        return 0
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    generator.load_state_dict(ckpt["generator"], strict=True)
    discriminator.load_state_dict(ckpt["discriminator"], strict=True)
    for key, obj in (("opt_g", opt_g), ("opt_d", opt_d), ("style_critic", style_critic), ("opt_sc", opt_sc),
                     ("frozen_encoder_model", frozen_encoder_model), ("opt_style_thaw", opt_style_thaw)):
        if obj is not None and key in ckpt:
            obj.load_state_dict(ckpt[key])
    return int(ckpt.get("epoch", 0))


def train_stage1(generator, discriminator, frozen_encoder, train_loader, cond_bank: torch.Tensor, device, stage_cfg: StageTrainConfig,
                 checkpoint_path: Path, resume: bool = False, dp=None) -> List[Dict]:
    """reference lab3_train.py:342-483: same arguments, same per-epoch history rows (nine loss means, stage, epoch, d-step counts).
    dp (gan_amd.dp.DataParallel, one process per GPU, `train_loader` yielding this rank's shard): rank 0's parameters, moments and resume
    epoch go to every rank, gradients are all-reduced per optimizer step, the history rows are means over the global batch on every rank,
    and rank 0 alone writes the checkpoint.  BatchNorm buffers stay per rank (rank 0's are the ones saved)."""
    generator.train()
    discriminator.train()
    cond_bank = cond_bank.to(device)
    opt_g = FusedAdamW(generator.parameters(), lr=stage_cfg.lr_g, betas=stage_cfg.betas, weight_decay=0.0)
    opt_d = FusedAdamW(discriminator.parameters(), lr=stage_cfg.lr_d, betas=stage_cfg.betas, weight_decay=0.0)
    start_epoch = _load_stage_checkpoint(checkpoint_path, generator, discriminator, opt_g, opt_d) if resume else 0
    if dp is not None:
        start_epoch = dp.broadcast_state(start_epoch, opt_g, opt_d)
    keys = ("loss_g", "loss_d", "loss_r1", "loss_adv", "loss_l1", "loss_content", "loss_style", "loss_continuity", "loss_mrstft")
    history: List[Dict] = []
    step_idx = 0
    # on unless stage_cfg.extra["hip_graph"] is False: the iteration replayed from a recorded HIP graph (same launches, same results to
    # run-to-run noise: tests/test_gpu_melgraph.py; twice the throughput at the reference's default batch of 32)
    graphed = GraphedStage1(stage_cfg, generator, discriminator, opt_g, opt_d, frozen_encoder) if stage_cfg.extra.get("hip_graph", True) else None
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
            if graphed is not None:
                out = graphed(real, zc, cond, step_idx, zs=zs, dp=dp)
            else:
                out = mel_stage1_iteration(stage_cfg, generator, discriminator, opt_g, opt_d, real, zc, cond, step_idx, zs=zs,
                                           frozen_encoder=frozen_encoder, dp=dp)
            sums += torch.stack([out[k].float().reshape(()) for k in keys])  # one host read per epoch, not nine per iteration
            taken += int((step_idx % int(max(1, stage_cfg.d_step_period))) == 0)
            nb += 1
            step_idx += 1
        n_terms = float(max(1, nb))
        if dp is not None and dp.world_size > 1:  # loss sums and iteration counts over all ranks: the row is the global batch's mean
            host = {k: float(v) for k, v in zip(keys, sums.tolist())}
            counts = {"nb": float(nb)}
            dp.reduce_history_means(host, counts)
            sums = torch.tensor([host[k] for k in keys])
            n_terms = max(1.0, counts["nb"])
        means = (sums / n_terms).tolist()
        ep: Dict = {k: float(v) for k, v in zip(keys, means)}
        ep.update(stage="stage1", epoch=int(epoch + 1), d_steps_taken=float(taken), d_steps_skipped=float(nb - taken),
                  d_step_rate=float(taken / max(1, nb)))
        history.append(ep)
        if dp is None or dp.rank == 0:
            _save_stage_checkpoint(checkpoint_path, int(epoch + 1), generator, discriminator, opt_g, opt_d, meta={"stage": "stage1"})
            print(f"[stage1] epoch={epoch + 1} loss_g={ep['loss_g']:.4f} loss_d={ep['loss_d']:.4f}")
    return history


STAGE2_KEYS = ("loss_g", "loss_d", "loss_r1", "loss_sc", "loss_adv", "loss_l1", "loss_continuity", "loss_mrstft", "loss_content",
               "loss_style_ce", "loss_style_mid", "loss_pfm_style", "loss_contrastive", "loss_batch_infonce_diversity", "loss_style_hinge",
               "loss_lowmid_recon", "loss_flatness", "loss_timbre_balance", "loss_spectral_tilt", "loss_zcr_proxy", "loss_hf_muzzle",
               "loss_highpass_anchor", "loss_feature_match", "loss_perceptual", "loss_diversity", "loss_mel_diversity",
               "loss_target_profile", "content_weight", "lr_g", "style_conf", "style_thaw_active", "cond_alpha")  # history row, :690-723


def _grad_if(weight) -> "contextlib.AbstractContextManager":
    """A term whose weight is exactly zero still goes into the history row (the reference logs all of them), but its gradient would be
    multiplied by zero: computed under no_grad, the value is the same, the generator's gradient is the same (the zero contributions are
    simply not added) and its backward pass -- for `style_mid` a whole encoder pass -- is not run.  The reference's CLI defaults leave 13
    of the 23 terms at zero (run_lab3.py:402-433)."""
    return contextlib.nullcontext() if float(weight) != 0.0 else torch.no_grad()


def _same_target_pair_hinge(feat: torch.Tensor, tgt_idx: torch.Tensor, margin: float, max_pairs: int) -> torch.Tensor:
    """`_diversity_loss` / `_mel_diversity_loss` (reference :580-605, :634-659): within every target genre, neighbours of a random
    permutation form pairs; mean over genres of mean relu(cos - margin), spending at most max_pairs pairs."""
    zero = torch.zeros((), device=feat.device)
    if feat.shape[0] < 2:
        return zero
    f = F.normalize(feat, dim=-1)
    budget, terms = int(max_pairs), []
    for g in torch.unique(tgt_idx):
        ids = torch.where(tgt_idx == g)[0]
        if ids.numel() < 2:
            continue
        order = ids[torch.randperm(ids.numel(), device=feat.device)]
        a, b = order[:-1], order[1:]
        if budget > 0 and a.numel() > budget:
            a, b = a[:budget], b[:budget]
        terms.append(F.relu((f[a] * f[b]).sum(dim=-1) - float(margin)).mean())
        budget -= int(a.numel())
        if budget <= 0:
            break
    return torch.stack(terms).mean() if terms else zero


def _within_genre_infonce(z_fake: torch.Tensor, tgt_idx: torch.Tensor, temp: float) -> torch.Tensor:
    """`_batch_infonce_diversity_loss` (reference :607-632): per target genre, cross-entropy of z z_detached^T / temp against the
    diagonal -- the other samples of the genre are the negatives."""
    zero = torch.zeros((), device=z_fake.device)
    if z_fake.shape[0] < 2:
        return zero
    z = F.normalize(z_fake, dim=-1)
    terms = []
    for g in torch.unique(tgt_idx):
        ids = torch.where(tgt_idx == g)[0]
        if ids.numel() < 2:
            continue
        zg = z[ids]
        terms.append(F.cross_entropy(zg @ zg.detach().t() / float(temp), torch.arange(zg.shape[0], device=z.device)))
    return torch.stack(terms).mean() if terms else zero


def _style_thaw_parameters(encoder_model, scope: str) -> list:
    """reference :515-529."""
    key = str(scope).strip().lower()
    parts = {"style_head": ("style_head",), "shared_style": ("shared", "style_head"), "full_style_path": ("backbone", "shared", "style_head")}
    if key not in parts:
        raise ValueError(f"Unsupported style_thaw_scope: {scope}")
    return [p for name in parts[key] for p in getattr(encoder_model, name).parameters()]


def _stage2_condition(stage_cfg: StageTrainConfig, cond_bank, exemplars: Dict[int, torch.Tensor], tgt_idx, epoch: int):
    """(cond [B, cond_dim], cond_alpha) of reference :739-791: target centroid, a drawn exemplar of the target genre written over its
    leading dimensions, or their scheduled mix; then conditioning noise / style jitter."""
    centroid = cond_bank[tgt_idx]
    mode = str(stage_cfg.cond_mode).strip().lower()
    exemplar = centroid
    if mode in {"exemplar", "mix"} and exemplars:
        rows = []
        for j, tg in enumerate(tgt_idx.tolist()):
            pool = exemplars.get(int(tg))
            if pool is None or pool.shape[0] == 0:
                rows.append(centroid[j])
                continue
            v = pool[int(torch.randint(0, int(pool.shape[0]), (1,), device=tgt_idx.device).item())]
            if float(stage_cfg.cond_exemplar_noise_std) > 0.0:
                v = F.normalize(v + torch.randn_like(v) * float(stage_cfg.cond_exemplar_noise_std), dim=-1)
            rows.append(v)
        raw = torch.stack(rows, dim=0)
        k = min(int(centroid.shape[1]), int(raw.shape[1]))  # a longer cond keeps the centroid's tail
        exemplar = F.normalize(torch.cat([raw[:, :k], centroid[:, k:]], dim=1), dim=-1)
    if mode == "exemplar":
        cond, alpha = exemplar, 0.0
    elif mode == "mix":
        alpha = _linear_weight(float(stage_cfg.cond_alpha_start), float(stage_cfg.cond_alpha_end), int(epoch), int(stage_cfg.epochs))
        cond = F.normalize(float(alpha) * centroid + (1.0 - float(alpha)) * exemplar, dim=-1)
    else:
        cond, alpha = centroid, 1.0
    noise = [torch.randn_like(cond) * float(std) for std in (stage_cfg.cond_noise_std, stage_cfg.style_jitter_std) if float(std) > 0.0]
    if noise:
        cond = F.normalize(cond + sum(noise), dim=-1)
    return cond, float(alpha)


def _target_reference_rows(gidx: torch.Tensor, tgt_idx: torch.Tensor) -> torch.Tensor:
    """reference :891-905: for every sample, a batch row whose genre is the sample's target (any row when there is none)."""
    src, n = gidx.tolist(), int(gidx.shape[0])
    by_genre: Dict[int, List[int]] = {}
    for j, g in enumerate(src):
        by_genre.setdefault(int(g), []).append(j)
    rows = []
    for tg in tgt_idx.tolist():
        cands = by_genre.get(int(tg), [])
        pick = int(torch.randint(0, len(cands) if cands else n, (1,), device=gidx.device).item())
        rows.append(cands[pick] if cands else pick)
    return torch.tensor(rows, device=gidx.device, dtype=torch.long)


def train_stage2(generator, discriminator, frozen_encoder, train_loader, cond_bank: torch.Tensor, style_proto_bank: torch.Tensor,
                 style_exemplar_bank: Optional[Dict[int, torch.Tensor]], target_profile_bank: Optional[torch.Tensor], device,
                 stage_cfg: StageTrainConfig, checkpoint_path: Path, resume: bool = False) -> List[Dict]:
    """reference lab3_train.py:486-1088: same arguments, same history rows, same checkpoint keys.  The generator and discriminator
    passes run on the HIP kernels; the frozen encoder is the caller's module; the 20-odd scalar penalties are device tensor expressions
    (lab3_losses.py).  Host reads per iteration: the D loss when `d_loss_floor_for_step` is set and the two confidences when
    `adaptive_content_weight` is -- both are data-dependent control flow in the reference; the history sums stay on the device."""
    generator.train()
    discriminator.train()
    cond_bank = cond_bank.to(device)
    protos = F.normalize(style_proto_bank.to(device), dim=-1)
    exemplars = {int(g): F.normalize(t.to(device).float(), dim=-1) for g, t in (style_exemplar_bank or {}).items()}
    if target_profile_bank is not None:
        target_profile_bank = target_profile_bank.to(device)
    n_genres = int(cond_bank.shape[0])
    style_critic = StyleCritic(in_dim=int(protos.shape[1]), n_classes=n_genres).to(device)
    style_critic.train()
    thaw_epochs = max(0, int(stage_cfg.style_thaw_last_epochs))
    thaw_params = _style_thaw_parameters(frozen_encoder.model, stage_cfg.style_thaw_scope)
    for p in thaw_params:
        p.requires_grad = False
    opt_thaw = torch.optim.Adam(thaw_params, lr=float(stage_cfg.style_thaw_lr), betas=stage_cfg.betas) if thaw_params else None
    opt_g = FusedAdamW(generator.parameters(), lr=stage_cfg.lr_g, betas=stage_cfg.betas, weight_decay=0.0)
    opt_d = FusedAdamW(discriminator.parameters(), lr=stage_cfg.lr_d, betas=stage_cfg.betas, weight_decay=0.0)
    opt_sc = FusedAdamW(style_critic.parameters(), lr=float(stage_cfg.style_critic_lr), betas=stage_cfg.betas, weight_decay=0.0)
    class_w = torch.ones((n_genres,), device=device, dtype=torch.float32)
    labels = getattr(getattr(train_loader, "dataset", None), "genre_idx", None)
    if bool(stage_cfg.style_class_balance) and labels is not None:  # inverse frequency, mean 1 (:542-549)
        inv = 1.0 / np.maximum(np.bincount(np.asarray(labels, dtype=np.int64), minlength=n_genres).astype(np.float32), 1.0)
        class_w = torch.from_numpy(inv / max(float(inv.mean()), 1e-8)).to(device=device, dtype=torch.float32)
    start_epoch = _load_stage_checkpoint(checkpoint_path, generator, discriminator, opt_g, opt_d, style_critic=style_critic, opt_sc=opt_sc,
                                         frozen_encoder_model=frozen_encoder.model, opt_style_thaw=opt_thaw) if resume else 0
    w = stage_cfg.weights
    smoothing = float(stage_cfg.style_label_smoothing)
    d_params = list(discriminator.parameters())
    history: List[Dict] = []
    step_idx = 0
    lp_bins: Optional[int] = None
    for epoch in range(start_epoch, int(stage_cfg.epochs)):
        thaw_active = bool(thaw_epochs > 0 and epoch >= int(stage_cfg.epochs) - thaw_epochs)
        for p in thaw_params:
            p.requires_grad = thaw_active
        warm = max(0, int(stage_cfg.g_lr_warmup_epochs))
        cur_lr_g = float(stage_cfg.lr_g)
        if warm > 0 and epoch < warm:  # linear from g_lr_start_mult to 1 over the warm-up epochs (:668-674)
            m0 = float(stage_cfg.g_lr_start_mult)
            cur_lr_g *= m0 + (1.0 - m0) * (float(epoch + 1) / float(warm))
        opt_g.lr = cur_lr_g
        style_only = max(0, int(stage_cfg.style_only_warmup_epochs))
        if epoch < style_only:
            content_w = 0.0
        elif stage_cfg.content_weight_start is not None and stage_cfg.content_weight_end is not None:
            content_w = _linear_weight(float(stage_cfg.content_weight_start), float(stage_cfg.content_weight_end), int(epoch) - style_only,
                                       max(1, int(stage_cfg.epochs) - style_only))
        else:
            content_w = float(w.content)
        sums = torch.zeros(len(STAGE2_KEYS), device=device, dtype=torch.float64)
        nb = taken = skipped = 0
        ema_conf: Optional[float] = None
        for bidx, batch in enumerate(train_loader):
            if stage_cfg.max_batches_per_epoch is not None and bidx >= int(stage_cfg.max_batches_per_epoch):
                break
            real = batch["mel_norm"].to(device).float()
            zc = batch["z_content"].to(device).float()
            gidx = batch["genre_idx"].to(device).long()
            tgt_idx = (_sample_shift_targets_balanced(gidx, n_genres, offset=step_idx) if bool(stage_cfg.target_balance)
                       else _sample_shift_targets(gidx, n_genres))
            cond, cond_alpha = _stage2_condition(stage_cfg, cond_bank, exemplars, tgt_idx, epoch)
            cond_d = torch.cat([cond, zc], dim=1) if bool(stage_cfg.disc_use_content_cond) else cond
            do_d_step = (step_idx % int(max(1, stage_cfg.d_step_period))) == 0
            real_db = denormalize_log_mel(real)
            if lp_bins is None:
                cut = stage_cfg.style_lowpass_cutoff_hz
                lp_bins = (LL.estimate_lowpass_keep_bins(int(real_db.shape[1]), int(frozen_encoder.cfg.sample_rate), float(cut))
                           if cut is not None and float(cut) > 0.0 else int(stage_cfg.style_lowpass_keep_bins))
            real_db_lp = LL.lowpass_mel_db(real_db, keep_bins=lp_bins)
            # style critic on the real latents (:806-814)
            with torch.no_grad():
                enc_real = frozen_encoder.forward_log_mel_tensor(real_db_lp)
            loss_sc = F.cross_entropy(style_critic(enc_real["z_style"].detach()), gidx, weight=class_w)
            _backward_step(loss_sc, opt_sc, 0.0)
            # D step (:816-840)
            with torch.no_grad():
                fake_det = generator(zc, cond)
            loss_d = _gan_d_loss(stage_cfg.gan_loss, discriminator(real, cond_d), discriminator(fake_det, cond_d),
                                 float(stage_cfg.d_real_label), float(stage_cfg.d_fake_label))
            r1 = torch.zeros((), device=device)
            if do_d_step and float(stage_cfg.r1_gamma) > 0.0 and (step_idx % int(max(1, stage_cfg.r1_interval)) == 0):
                r1 = 0.5 * float(stage_cfg.r1_gamma) * _r1_penalty(discriminator, real, cond_d)
                loss_d = loss_d + r1
            floor = float(stage_cfg.d_loss_floor_for_step)
            if do_d_step and (floor <= 0.0 or float(loss_d.item()) >= floor):
                _backward_step(loss_d, opt_d, stage_cfg.d_grad_clip_norm)
                taken += 1
            else:
                skipped += 1
            # G step (:842-1028)
            fake = generator(zc, cond)
            saved = [p.requires_grad for p in d_params]
            for p in d_params:  # the reference lets D's weight gradients of this pass accumulate and then discards them
                p.requires_grad_(False)
            try:
                logit_fake_g, d_fake_feat = discriminator(fake, cond_d, return_features=True)
            finally:
                for p, r in zip(d_params, saved):
                    p.requires_grad_(r)
            loss_adv = _gan_g_loss(stage_cfg.gan_loss, logit_fake_g, float(stage_cfg.g_real_label), float(stage_cfg.d_fake_label))
            with torch.no_grad():
                _, d_real_feat = discriminator(real, cond_d, return_features=True)
            # (every term is computed for the history row; the ones weighted zero without a backward pass: _grad_if)
            with _grad_if(w.feature_match):
                loss_fm = F.l1_loss(d_fake_feat, d_real_feat)
            fake_db = denormalize_log_mel(fake)
            enc_out = frozen_encoder.forward_log_mel_tensor(LL.lowpass_mel_db(fake_db, keep_bins=lp_bins))
            loss_content = LL.cosine_distance(enc_out["z_content"], zc).mean()
            fake_zs = enc_out["z_style"]
            style_logits = style_critic(fake_zs)
            loss_style = F.cross_entropy(style_logits, tgt_idx, weight=class_w, label_smoothing=smoothing)
            conf = torch.softmax(style_logits, dim=1).gather(1, tgt_idx.unsqueeze(1)).squeeze(1)
            style_conf = conf.mean()
            with _grad_if(w.style_mid):
                enc_mid = frozen_encoder.forward_log_mel_tensor(LL.bandpass_mel_db(fake_db, low_bin=int(stage_cfg.style_mid_low_bin),
                                                                                  high_bin=int(stage_cfg.style_mid_high_bin)))
                loss_style_mid = F.cross_entropy(style_critic(enc_mid["z_style"]), tgt_idx, weight=class_w, label_smoothing=smoothing)
            with _grad_if(w.style_hinge):
                loss_style_hinge = F.relu(float(stage_cfg.style_hinge_target_conf) - conf).mean()
            with _grad_if(w.contrastive):
                proto_logits = F.normalize(fake_zs, dim=-1) @ protos.t() / float(stage_cfg.contrastive_temp)  # :572-578
                loss_contrastive = F.cross_entropy(proto_logits, tgt_idx)
                contrastive_conf = torch.softmax(proto_logits, dim=1).gather(1, tgt_idx.unsqueeze(1)).squeeze(1).mean()
            with torch.no_grad():
                enc_ref = frozen_encoder.forward_log_mel_tensor(real_db_lp[_target_reference_rows(gidx, tgt_idx)])
            with _grad_if(w.pfm_style):
                loss_pfm = (F.mse_loss(F.normalize(enc_out["shared_feat"], dim=-1), F.normalize(enc_ref["shared_feat"].detach(), dim=-1))
                            + F.mse_loss(F.normalize(fake_zs, dim=-1), F.normalize(enc_ref["z_style"].detach(), dim=-1)))
            with _grad_if(w.batch_infonce_diversity):
                loss_infonce = _within_genre_infonce(fake_zs, tgt_idx, float(stage_cfg.batch_infonce_temp))
            loss_l1 = F.l1_loss(fake, real)
            with _grad_if(float(w.continuity) if float(w.continuity) != 0.0 else float(w.mrstft)):
                loss_continuity = multi_resolution_stft_loss(fake, real, resolutions=stage_cfg.mrstft_resolutions,
                                                             precision=getattr(generator, "precision", "fp32"))
            loss_mrstft = loss_continuity
            with _grad_if(w.lowmid_recon):
                loss_lowmid = LL.frequency_weighted_l1_loss(fake, real, split_bin=int(stage_cfg.lowmid_split_bin),
                                                            low_gain=float(stage_cfg.lowmid_gain), high_gain=float(stage_cfg.high_gain))
            with _grad_if(w.highpass_anchor):
                loss_highpass = LL.highpass_recon_l1_loss(fake, real, low_cut_bin=int(stage_cfg.highpass_cut_bin), gain=float(stage_cfg.highpass_gain))
            with _grad_if(w.flatness):
                loss_flat = LL.spectral_flatness_penalty_from_db(fake_db)
            with _grad_if(w.timbre_balance):
                loss_timbre = LL.timbre_balance_penalty_from_db(fake_db)
            with _grad_if(w.spectral_tilt):
                loss_tilt = LL.spectral_tilt_penalty_from_db(fake_db, hf_to_lf_max_ratio=float(stage_cfg.spectral_tilt_max_ratio))
            with _grad_if(w.zcr_proxy):
                loss_zcr = LL.zcr_proxy_penalty_from_db(fake_db, target_max=float(stage_cfg.zcr_proxy_target_max))
            with _grad_if(w.hf_muzzle):
                loss_hf = LL.hf_excess_penalty_from_db(fake_db, real_db, top_frac=float(stage_cfg.hf_top_frac),
                                                       max_ratio_multiplier=float(stage_cfg.hf_max_ratio_multiplier))
            with _grad_if(w.perceptual):
                loss_perc = F.l1_loss(enc_out["shared_feat"], enc_real["shared_feat"].detach())
            with _grad_if(w.diversity):
                loss_div = _same_target_pair_hinge(d_fake_feat, tgt_idx, float(stage_cfg.diversity_margin), int(stage_cfg.diversity_max_pairs))
            with _grad_if(w.mel_diversity):
                loss_mel_div = _same_target_pair_hinge(fake.flatten(1), tgt_idx, float(stage_cfg.mel_diversity_margin),
                                                       int(stage_cfg.mel_diversity_max_pairs))
            if target_profile_bank is not None:  # time-summed power per mel bin, as a distribution (:971-976)
                with _grad_if(w.target_profile):
                    prof = torch.pow(10.0, fake_db / 10.0).clamp_min(1e-10).sum(dim=2)
                    loss_prof = F.l1_loss(prof / (prof.sum(dim=1, keepdim=True) + 1e-8), target_profile_bank[tgt_idx])
            else:
                loss_prof = torch.zeros((), device=fake.device)
            if stage_cfg.adaptive_content_weight:  # :980-990
                cur = float((0.5 * style_conf + 0.5 * contrastive_conf).item())
                ema_conf = cur if ema_conf is None else 0.9 * ema_conf + 0.1 * cur
                if ema_conf < float(stage_cfg.adaptive_conf_low):
                    content_w = float(stage_cfg.adaptive_content_low)
                elif ema_conf > float(stage_cfg.adaptive_conf_high):
                    content_w = float(stage_cfg.adaptive_content_high)
            loss_g = (w.adv * loss_adv + w.recon_l1 * loss_l1 + w.continuity * loss_continuity + float(content_w) * loss_content
                      + w.style * loss_style + w.style_mid * loss_style_mid + w.pfm_style * loss_pfm + w.style_hinge * loss_style_hinge
                      + w.contrastive * loss_contrastive + w.batch_infonce_diversity * loss_infonce + w.mrstft * loss_mrstft
                      + w.lowmid_recon * loss_lowmid + w.highpass_anchor * loss_highpass + w.flatness * loss_flat
                      + w.timbre_balance * loss_timbre + w.spectral_tilt * loss_tilt + w.zcr_proxy * loss_zcr + w.hf_muzzle * loss_hf
                      + w.feature_match * loss_fm + w.perceptual * loss_perc + w.diversity * loss_div + w.mel_diversity * loss_mel_div
                      + w.target_profile * loss_prof)
            opt_g.zero_grad(set_to_none=True)
            style_critic.zero_grad(set_to_none=True)  # the critic is only an evaluator here; its own step zeroes before its backward
            if thaw_active and opt_thaw is not None:
                opt_thaw.zero_grad(set_to_none=True)
            loss_g.backward()
            clip = float(stage_cfg.g_grad_clip_norm)
            opt_g.max_grad_norm = clip if clip > 0.0 else 0.0
            opt_g.step()
            if thaw_active and opt_thaw is not None:
                if clip > 0.0 and thaw_params:
                    torch.nn.utils.clip_grad_norm_(thaw_params, max_norm=clip)
                opt_thaw.step()
            row = (loss_g, loss_d, r1, loss_sc, loss_adv, loss_l1, loss_continuity, loss_mrstft, loss_content, loss_style, loss_style_mid,
                   loss_pfm, loss_contrastive, loss_infonce, loss_style_hinge, loss_lowmid, loss_flat, loss_timbre, loss_tilt, loss_zcr, loss_hf,
                   loss_highpass, loss_fm, loss_perc, loss_div, loss_mel_div, loss_prof)
            host = torch.tensor([float(content_w), float(cur_lr_g), 0.0, 1.0 if thaw_active else 0.0, float(cond_alpha)], dtype=torch.float64)
            vals = torch.cat([torch.stack([v.detach().float().reshape(()) for v in row]).double(), host.to(device)])  # (one cast, not 27)
            vals[STAGE2_KEYS.index("style_conf")] = style_conf.detach().double()
            sums += vals
            nb += 1
            step_idx += 1
        means = (sums / nb).tolist() if nb > 0 else [float("nan")] * len(STAGE2_KEYS)
        ep: Dict = {k: float(v) for k, v in zip(STAGE2_KEYS, means)}
        ep.update(stage="stage2", epoch=int(epoch + 1), d_steps_taken=float(taken), d_steps_skipped=float(skipped),
                  d_step_rate=float(taken / max(1, taken + skipped)))
        history.append(ep)
        _save_stage_checkpoint(checkpoint_path, int(epoch + 1), generator, discriminator, opt_g, opt_d, meta={"stage": "stage2"},
                               style_critic=style_critic, opt_sc=opt_sc, frozen_encoder_model=frozen_encoder.model, opt_style_thaw=opt_thaw)
        print(f"[stage2] epoch={epoch + 1} loss_g={ep['loss_g']:.4f} loss_d={ep['loss_d']:.4f}")
    return history
