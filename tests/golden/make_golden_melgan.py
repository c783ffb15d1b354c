#!/usr/bin/env python3
"""Golden vectors for the MEL-TRACK GAN pieces, FROM THE REFERENCE'S OWN CLASSES.

Run in the build container only:   python tests/golden/make_golden_melgan.py      # writes tests/golden/melgan_golden.npz

Imports /root/reference/lab 3/src/lab3_models.py (torch-only) and lab3_train.py (its GAN loss helpers; imported with empty stub
modules for the absent torchaudio / librosa / soundfile, as make_golden.py does) and records, on procedurally filled parameters and
inputs (oracle/procfill.py; no weight file is committed):
  * ReconstructionDecoder forward for transpose / pixelshuffle / nearest up-sampling, instance / batch norm, with and without MRF
    blocks, and the gradient fingerprints of every parameter for the default configuration;
  * MelDiscriminator (with and without the projection head), MultiScale / SubBand / MultiPeriod / Hybrid wrappers: logits, pooled
    features, BatchNorm buffers after the pass, parameter-gradient fingerprints;
  * bce_logits, _gan_d_loss, _gan_g_loss (bce and hinge, smoothed labels);
  * three consecutive iterations of train_stage1's loop body (lab3_train.py:380-456: D step, G step, Adam) assembled from the
    imported pieces, content / style terms off (they need the caller's Lab1 encoder): losses, gradient fingerprints of the first
    iteration, parameter fingerprints and BatchNorm buffers after each.
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/lab 3")

for _name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional", "librosa", "soundfile"):
    if _name not in sys.modules:
        sys.modules[_name] = types.ModuleType(_name)

from src import lab3_models as RM  # noqa: E402
from src import lab3_train as RT  # noqa: E402
from src import lab3_losses as RL  # noqa: E402

from oracle.procfill import fill_state_dict, proc_input  # noqa: E402

torch.set_num_threads(8)
STORE = {}


def fp(t: torch.Tensor) -> np.ndarray:
    """(sum, abs-sum, sum of squares, strided samples) of a tensor: a fingerprint that pins it to ~1e-6 without storing it."""
    d = t.detach().double().reshape(-1)
    step = max(1, d.numel() // 8)
    return np.concatenate([[d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], d[::step][:8].numpy()]).astype(np.float64)


def put(name, t):
    STORE[name] = t.detach().numpy().copy() if torch.is_tensor(t) else np.asarray(t)


def filled(mod: torch.nn.Module, seed: int):
    mod.load_state_dict(fill_state_dict(mod.state_dict(), seed))
    return mod


def decoder_cases():
    B = 2
    for tag, kw in (("default", {}), ("ps_batch", dict(upsample="pixelshuffle", norm="batch")), ("nearest", dict(upsample="nearest")),
                    ("mrf", dict(mrf=True, base_channels=64)), ("small", dict(zc_dim=16, cond_dim=8, n_mels=16, n_frames=32, base_channels=32))):
        G = filled(RM.ReconstructionDecoder(**kw), 11).train()
        zc = proc_input((B, G.fc.in_features - kw.get("cond_dim", 160)), f"melgan.zc.{tag}", 0)
        v = proc_input((B, kw.get("cond_dim", 160)), f"melgan.v.{tag}", 0)
        mel = G(zc, v)
        put(f"dec/{tag}/out_fp", fp(mel))
        put(f"dec/{tag}/out_slice", mel[:, ::5, ::17])
        if tag in ("default", "small", "ps_batch"):
            tgt = proc_input(tuple(mel.shape), f"melgan.tgt.{tag}", 0)
            loss = ((mel - tgt) ** 2).mean() + 0.1 * mel.abs().mean()
            put(f"dec/{tag}/loss", loss)
            names = [n for n, _ in G.named_parameters()]
            grads = torch.autograd.grad(loss, [p for _, p in G.named_parameters()])
            for n, g in zip(names, grads):
                put(f"dec/{tag}/grad/{n}", fp(g))
            if tag == "ps_batch":
                for n, b in G.named_buffers():
                    put(f"dec/{tag}/buf/{n}", b)


def disc_cases():
    B = 3
    mel = proc_input((B, 96, 256), "melgan.d.mel", 0, 0.8)
    cond = proc_input((B, 160), "melgan.d.cond", 0)
    for tag, mk, use_cond in (("plain", lambda: RM.MelDiscriminator(cond_dim=0), False),
                              ("proj", lambda: RM.MelDiscriminator(cond_dim=160), True),
                              ("proj_small", lambda: RM.MelDiscriminator(base_channels=16, cond_dim=160), True),
                              ("multiscale", lambda: RM.MultiScaleMelDiscriminator(cond_dim=160, base_channels=16), True),
                              ("subband", lambda: RM.SubBandMelDiscriminator(cond_dim=160, base_channels=16), True),
                              ("multiperiod", lambda: RM.MultiPeriodMelDiscriminator(cond_dim=160, base_channels=16), True),
                              ("hybrid", lambda: RM.HybridMelDiscriminator(cond_dim=160), True)):
        D = filled(mk(), 12).train()
        x = mel.clone().requires_grad_(True)
        logit, feat = D(x, cond if use_cond else None, return_features=True)
        put(f"disc/{tag}/logit", logit)
        put(f"disc/{tag}/feat_fp", fp(feat))
        loss = F.softplus(-logit).mean() + 0.01 * feat.pow(2).mean()
        put(f"disc/{tag}/loss", loss)
        names = [n for n, _ in D.named_parameters()]
        grads = torch.autograd.grad(loss, [x] + [p for _, p in D.named_parameters()])
        put(f"disc/{tag}/grad_x_fp", fp(grads[0]))
        put(f"disc/{tag}/grad_x_slice", grads[0][:, ::9, ::31])
        for n, g in zip(names, grads[1:]):
            put(f"disc/{tag}/grad/{n}", fp(g))
        for n, b in D.named_buffers():
            if b.dtype.is_floating_point:
                put(f"disc/{tag}/buf/{n}", fp(b))
        if tag == "proj":
            D.eval()
            put(f"disc/{tag}/eval_logit", D(mel, cond))


def loss_cases():
    a = proc_input((7,), "melgan.loss.a", 0, 2.0)
    b = proc_input((7,), "melgan.loss.b", 0, 2.0)
    put("loss/bce_real", RM.bce_logits(a, True))
    put("loss/bce_fake_smooth", RM.bce_logits(a, False, real_label=0.9, fake_label=0.1))
    for key in ("bce", "hinge"):
        put(f"loss/d_{key}", RT._gan_d_loss(key, a, b, 0.9, 0.05))
        put(f"loss/g_{key}", RT._gan_g_loss(key, b, 0.95, 0.05))


def iteration_sequence():
    """train_stage1's loop body (lab3_train.py:380-456) three times over, gan_loss 'bce' then 'hinge' with clipping."""
    for tag, gan_loss, clip in (("bce", "bce", 0.0), ("hinge_clip", "hinge", 1.5)):
        G = filled(RM.ReconstructionDecoder(zc_dim=32, cond_dim=24, n_mels=32, n_frames=64, base_channels=64), 21).train()
        D = filled(RM.MelDiscriminator(base_channels=16, cond_dim=24), 22).train()
        cfg = RT.StageTrainConfig(stage_name="stage1", epochs=1, gan_loss=gan_loss, g_grad_clip_norm=clip, d_grad_clip_norm=clip,
                                  mrstft_resolutions=((16, 4, 16), (32, 8, 32), (64, 16, 64)))
        opt_g = torch.optim.Adam(G.parameters(), lr=cfg.lr_g, betas=cfg.betas)
        opt_d = torch.optim.Adam(D.parameters(), lr=cfg.lr_d, betas=cfg.betas)
        for it in range(3):
            real = proc_input((4, 32, 64), f"melgan.it{it}.real", 0, 0.8)
            zc = proc_input((4, 32), f"melgan.it{it}.zc", 0)
            cond = proc_input((4, 24), f"melgan.it{it}.cond", 0)
            cond_d = cond
            with torch.no_grad():
                fake_det = G(zc, cond)
            loss_d = RT._gan_d_loss(cfg.gan_loss, D(real, cond_d), D(fake_det, cond_d), cfg.d_real_label, cfg.d_fake_label)
            opt_d.zero_grad(set_to_none=True)
            loss_d.backward()
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(D.parameters(), max_norm=clip)
            if it == 0:
                for n, p in D.named_parameters():
                    put(f"iter/{tag}/0/d_grad/{n}", fp(p.grad))
            opt_d.step()
            fake = G(zc, cond)
            loss_adv = RT._gan_g_loss(cfg.gan_loss, D(fake, cond_d), cfg.g_real_label, cfg.d_fake_label)
            loss_l1 = F.l1_loss(fake, real)
            loss_cont = RL.multi_resolution_stft_loss(fake, real, resolutions=cfg.mrstft_resolutions)
            w = cfg.weights
            loss_g = w.adv * loss_adv + w.recon_l1 * loss_l1 + w.continuity * loss_cont + w.mrstft * loss_cont
            opt_g.zero_grad(set_to_none=True)
            loss_g.backward()
            if clip > 0:
                torch.nn.utils.clip_grad_norm_(G.parameters(), max_norm=clip)
            if it == 0:
                for n, p in G.named_parameters():
                    put(f"iter/{tag}/0/g_grad/{n}", fp(p.grad))
            opt_g.step()
            for k, v in (("loss_d", loss_d), ("loss_g", loss_g), ("loss_adv", loss_adv), ("loss_l1", loss_l1), ("loss_continuity", loss_cont)):
                put(f"iter/{tag}/{it}/{k}", v)
            for n, p in list(G.named_parameters()) + [("D." + n, p) for n, p in D.named_parameters()]:
                put(f"iter/{tag}/{it}/param/{n}", fp(p))
            for n, b in D.named_buffers():
                if b.dtype.is_floating_point:
                    put(f"iter/{tag}/{it}/dbuf/{n}", b)


def main():
    decoder_cases()
    disc_cases()
    loss_cases()
    iteration_sequence()
    path = HERE / "melgan_golden.npz"
    np.savez_compressed(path, **STORE)
    print("wrote", path, len(STORE), "entries", path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
