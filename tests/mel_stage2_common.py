"""Shared set-up of the mel-track STAGE-2 parity case: used by tests/golden/make_golden_melstage2.py (which hands these objects to the
reference's own train_stage2, lab 3/src/lab3_train.py:486-1088, in the build container) and by tests/test_gpu_melstage2.py (which hands
the same objects to gan_amd.lab3_train.train_stage2 on the GPU).  Nothing here imports the reference.

Everything random in the reference's loop is taken out of the case, not out of the code under test:
  * targets are the balanced ones (`target_balance=True`: arange + step offset, :196-204);
  * `cond_mode="mix"` with no exemplar bank (:742-766 falls through to the centroid), noise and jitter 0;
  * every target genre has exactly two candidates in the batch and they carry the SAME real mel (rows j and j+3), so the reference
    row drawn at :897-905 is the same tensor whichever index the generator picks, and every same-target group of the diversity terms
    (:580-659) is one pair, whose similarity does not depend on the permutation;
  * StyleCritic's dropout (p=0.1, lab3_models.py:417-424) is set to 0 by the caller on both sides; its initial weights come from
    torch.manual_seed(STAGE2_SEED) on the CPU generator just before train_stage2 is called (the module is built on the CPU, then moved).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.procfill import fill_state_dict, proc_input

STAGE2_SEED = 4242
N_MELS, N_FRAMES, ZC_DIM, COND_DIM, N_GENRES, BATCH = 32, 64, 32, 24, 3, 6
G_KW = dict(zc_dim=ZC_DIM, cond_dim=COND_DIM, n_mels=N_MELS, n_frames=N_FRAMES, base_channels=64)


class _ToyEncoderNet(nn.Module):
    """A caller-side frozen encoder with the attribute layout train_stage2 touches (`backbone`, `shared`, `style_head`: the thaw
    scopes of :516-527) and the output keys it reads (`shared_feat`, `z_content`, `z_style`)."""

    def __init__(self, z_dim: int = ZC_DIM):
        super().__init__()
        self.backbone = nn.Sequential(nn.Conv2d(1, 8, 3, 2, 1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 16, 3, 2, 1), nn.BatchNorm2d(16), nn.ReLU(),
                                      nn.AdaptiveAvgPool2d((1, 1)))
        self.shared = nn.Linear(16, 48)
        self.content_head = nn.Linear(48, z_dim)
        self.style_head = nn.Linear(48, z_dim)

    def forward(self, log_mel: torch.Tensor) -> Dict[str, torch.Tensor]:
        h = self.backbone(log_mel.unsqueeze(1) * (1.0 / 40.0) + 1.0).flatten(1)
        h = F.relu(self.shared(h))
        return {"shared_feat": h, "z_content": F.normalize(self.content_head(h), dim=-1), "z_style": F.normalize(self.style_head(h), dim=-1)}


class ToyFrozenEncoder:
    """Stands in for FrozenLab1Encoder (lab3_bridge.py:159-234): `.model`, `.cfg.sample_rate`, `forward_log_mel_tensor`."""

    def __init__(self, device="cpu"):
        net = _ToyEncoderNet()
        net.load_state_dict(fill_state_dict(net.state_dict(), 31))
        self.model = net.to(device).eval()
        for p in self.model.parameters():
            p.requires_grad = False
        self.cfg = SimpleNamespace(sample_rate=22050, chunk_seconds=5.0, z_dim=ZC_DIM)

    def forward_log_mel_tensor(self, log_mel: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self.model(log_mel)


class ListLoader:
    """A DataLoader stand-in: iterates a fixed list of batches; `.dataset.genre_idx` feeds the class weights (:542-549)."""

    def __init__(self, batches: List[Dict[str, torch.Tensor]], genre_idx):
        self.batches = batches
        self.dataset = SimpleNamespace(genre_idx=np.asarray(genre_idx, dtype=np.int64))

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def stage2_batches(n: int) -> List[Dict[str, torch.Tensor]]:
    out = []
    for it in range(n):
        half = proc_input((BATCH // 2, N_MELS, N_FRAMES), f"melstage2.it{it}.real", 0, 0.8)
        zc = F.normalize(proc_input((BATCH, ZC_DIM), f"melstage2.it{it}.zc", 0), dim=-1)
        out.append({"mel_norm": torch.cat([half, half], 0), "z_content": zc, "z_style": F.normalize(proc_input((BATCH, ZC_DIM), f"melstage2.it{it}.zs", 0), dim=-1),
                    "genre_idx": torch.tensor([0, 1, 2, 0, 1, 2])})
    return out


def stage2_banks():
    cond_bank = F.normalize(proc_input((N_GENRES, COND_DIM), "melstage2.cond_bank", 0), dim=-1)
    style_proto = proc_input((N_GENRES, ZC_DIM), "melstage2.proto", 0)
    prof = proc_input((N_GENRES, N_MELS), "melstage2.profile", 0).abs() + 0.05
    return cond_bank, style_proto, prof / prof.sum(dim=1, keepdim=True)


ALL_WEIGHTS = dict(adv=0.5, recon_l1=8.0, content=2.0, style=1.0, continuity=1.0, mrstft=0.5, flatness=0.5, feature_match=2.0, perceptual=0.7,
                   style_hinge=0.6, contrastive=0.4, batch_infonce_diversity=0.3, pfm_style=1.5, diversity=0.8, timbre_balance=0.9,
                   lowmid_recon=0.6, spectral_tilt=0.7, zcr_proxy=1.1, hf_muzzle=1.3, highpass_anchor=0.45, style_mid=0.55, mel_diversity=0.65,
                   target_profile=3.0)

_BINS = dict(style_lowpass_keep_bins=24, style_mid_low_bin=4, style_mid_high_bin=20, lowmid_split_bin=20, highpass_cut_bin=6,
             mrstft_resolutions=((16, 4, 16), (32, 8, 32), (64, 16, 64)), cond_mode="mix", cond_noise_std=0.0, style_jitter_std=0.0,
             target_balance=True, diversity_margin=-0.5, mel_diversity_margin=0.2, zcr_proxy_target_max=0.01, spectral_tilt_max_ratio=0.3)

# name -> (StageTrainConfig keyword arguments without `weights`, number of batches in the loader, genre histogram of the data set)
CASES = {
    "all_terms": (dict(stage_name="stage2", epochs=3, max_batches_per_epoch=1, gan_loss="bce", g_lr_warmup_epochs=2, g_lr_start_mult=0.25,
                       content_weight_start=0.5, content_weight_end=2.0, style_only_warmup_epochs=1, style_label_smoothing=0.1,
                       style_class_balance=True, g_grad_clip_norm=2.0, d_grad_clip_norm=1.5, d_real_label=0.9, d_fake_label=0.05,
                       g_real_label=0.95, **_BINS), 3, [0, 0, 0, 1, 1, 2]),
    "hinge_thaw": (dict(stage_name="stage2", epochs=2, gan_loss="hinge", d_step_period=2, disc_use_content_cond=True, adaptive_content_weight=True,
                        adaptive_conf_low=0.9, adaptive_conf_high=0.95, d_loss_floor_for_step=0.2, style_thaw_last_epochs=1,
                        style_thaw_scope="shared_style", style_thaw_lr=1e-3, style_lowpass_cutoff_hz=6000.0, **_BINS), 2, [0, 1, 2]),
    "bce_r1": (dict(stage_name="stage2", epochs=1, gan_loss="bce", r1_gamma=1.5, r1_interval=2, d_loss_floor_for_step=0.5, **_BINS), 3, [0, 1, 2]),
}


def loader_for(case: str) -> ListLoader:
    kw, nb, hist = CASES[case]
    if case == "all_terms":  # one batch per epoch through max_batches_per_epoch: a different batch each epoch would need a re-iterable
        return ListLoader(stage2_batches(1), hist)
    return ListLoader(stage2_batches(nb), hist)

KEEP_BIN_ARGS = ((96, 22050, 4000.0), (96, 22050, 11025.0), (32, 22050, 6000.0), (1, 22050, 100.0), (64, 24000, 5.0), (80, 16000, 99999.0))
