#!/usr/bin/env python3
"""Golden vectors for the mel track's STAGE 2, FROM THE REFERENCE'S OWN train_stage2.

Run in the build container only:   python tests/golden/make_golden_melstage2.py      # writes tests/golden/melstage2_golden.npz

Imports /root/reference/lab 3/src/{lab3_models,lab3_train,lab3_losses}.py (empty stub modules for the absent torchaudio / librosa /
soundfile, as make_golden_melgan.py does) and records
  * every penalty of lab3_losses.py (:74-286) on a procedural dB mel: value and input-gradient fingerprint, the masks of
    lowpass_mel_db / bandpass_mel_db, estimate_lowpass_keep_bins on a few arguments;
  * the history rows train_stage2 (lab3_train.py:486-1088) returns for the two cases of tests/mel_stage2_common.py, and fingerprints
    of every tensor of the checkpoint it writes (generator, discriminator, style critic, the thawed encoder);
  * _r1_penalty (:329-339): value, parameter gradients, BatchNorm buffers; train_stage1 with R1 on;
  * ReconstructionDecoder / MelDiscriminator with spectral_norm=True: state-dict keys, two training passes, gradients, u / v buffers,
    an eval pass.
The only change made to the reference while it runs: StyleCritic is built with dropout 0 (see mel_stage2_common.py).
"""
from __future__ import annotations

import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, "/root/reference/lab 3")

for _name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional", "librosa", "soundfile"):
    if _name not in sys.modules:
        sys.modules[_name] = types.ModuleType(_name)

from src import lab3_models as RM  # noqa: E402
from src import lab3_train as RT  # noqa: E402
from src import lab3_losses as RL  # noqa: E402

import mel_stage2_common as C  # noqa: E402
from oracle.procfill import fill_state_dict, proc_input  # noqa: E402

torch.set_num_threads(8)
STORE = {}


def fp(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double().reshape(-1)
    step = max(1, d.numel() // 8)
    return np.concatenate([[d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], d[::step][:8].numpy()]).astype(np.float64)


def put(name, t):
    STORE[name] = t.detach().numpy().copy() if torch.is_tensor(t) else np.asarray(t)


def penalty_cases():
    db = proc_input((3, 32, 64), "melstage2.pen.db", 0, 40.0) - 40.0
    db[:, 20:, :] += 15.0  # bright upper bands so that the HF penalties are active
    db = db.clamp(-80.0, 0.0)
    db2 = (proc_input((3, 32, 64), "melstage2.pen.db2", 0, 40.0) - 40.0).clamp(-80.0, 0.0)
    norm_a = proc_input((3, 32, 64), "melstage2.pen.a", 0)
    norm_b = proc_input((3, 32, 64), "melstage2.pen.b", 0)
    put("pen/in_db_fp", fp(db))
    cases = {
        "flatness_closed": lambda x: RL.spectral_flatness_penalty_from_db(x),
        "timbre": lambda x: RL.timbre_balance_penalty_from_db(x, hf_max_ratio=0.2, lf_min_ratio=0.3),
        "tilt": lambda x: RL.spectral_tilt_penalty_from_db(x, hf_to_lf_max_ratio=0.3),
        "zcr": lambda x: RL.zcr_proxy_penalty_from_db(x, target_max=0.01),
        "hf_excess": lambda x: RL.hf_excess_penalty_from_db(x, db2, top_frac=0.25, max_ratio_multiplier=0.5),
    }
    for name, fn in cases.items():
        x = db.clone().requires_grad_(True)
        v = fn(x)
        put(f"pen/{name}/value", v)
        g, = torch.autograd.grad(v, x, allow_unused=True)
        put(f"pen/{name}/grad_fp", fp(g if g is not None else torch.zeros_like(x)))
    flat_db = proc_input((3, 32, 64), "melstage2.pen.flat", 0, 6.0) - 30.0  # a nearly white spectrum: flatness well above the hinge
    x = flat_db.clone().requires_grad_(True)
    v = RL.spectral_flatness_penalty_from_db(x)
    put("pen/flatness/value", v)
    put("pen/flatness/grad_fp", fp(torch.autograd.grad(v, x)[0]))
    for name, fn in {"freq_l1": lambda x: RL.frequency_weighted_l1_loss(x, norm_b, split_bin=20, low_gain=5.0, high_gain=0.5),
                     "highpass_l1": lambda x: RL.highpass_recon_l1_loss(x, norm_b, low_cut_bin=6, gain=2.0),
                     "cosine": lambda x: RL.cosine_distance(x[:, 0, :], norm_b[:, 0, :]).mean()}.items():
        x = norm_a.clone().requires_grad_(True)
        v = fn(x)
        put(f"pen/{name}/value", v)
        put(f"pen/{name}/grad_fp", fp(torch.autograd.grad(v, x)[0]))
    put("pen/lowpass_fp", fp(RL.lowpass_mel_db(db, keep_bins=24)))
    put("pen/lowpass_all_fp", fp(RL.lowpass_mel_db(db, keep_bins=400)))
    put("pen/bandpass_fp", fp(RL.bandpass_mel_db(db, low_bin=4, high_bin=20)))
    put("pen/bandpass_edge_fp", fp(RL.bandpass_mel_db(db, low_bin=0, high_bin=99)))
    put("pen/keep_bins", np.asarray([RL.estimate_lowpass_keep_bins(n, sr, c) for n, sr, c in C.KEEP_BIN_ARGS], dtype=np.int64))


def spectral_norm_cases():
    """spectral_norm=True (lab3_models.py:77-78, :152-153): two training-mode passes (u / v advance one power iteration each), the
    parameter gradients of the second, the u / v buffers after it, then an eval-mode pass (no power iteration)."""
    G = RM.ReconstructionDecoder(zc_dim=16, cond_dim=8, n_mels=16, n_frames=32, base_channels=32, spectral_norm=True)
    G.load_state_dict(fill_state_dict(G.state_dict(), 41))
    D = RM.MelDiscriminator(base_channels=16, cond_dim=8, spectral_norm=True)
    D.load_state_dict(fill_state_dict(D.state_dict(), 42))
    G.train(), D.train()
    put("sn/keys_g", np.asarray(list(G.state_dict().keys())))
    put("sn/keys_d", np.asarray(list(D.state_dict().keys())))
    zc, v = proc_input((3, 16), "melsn.zc", 0), proc_input((3, 8), "melsn.v", 0)
    mel_in = proc_input((3, 32, 64), "melsn.mel", 0, 0.8)
    for rep in (0, 1):
        mel = G(zc, v)
        logit, feat = D(mel_in, v, return_features=True)
        put(f"sn/{rep}/g_out_fp", fp(mel))
        put(f"sn/{rep}/d_logit", logit)
        put(f"sn/{rep}/d_feat_fp", fp(feat))
    loss_g = (mel ** 2).mean() + 0.1 * mel.abs().mean()
    for (n, _), g in zip(G.named_parameters(), torch.autograd.grad(loss_g, list(G.parameters()))):
        put(f"sn/g_grad/{n}", fp(g))
    loss_d = torch.nn.functional.softplus(-logit).mean() + 0.01 * feat.pow(2).mean()
    for (n, _), g in zip(D.named_parameters(), torch.autograd.grad(loss_d, list(D.parameters()))):
        put(f"sn/d_grad/{n}", fp(g))
    for tag, mod in (("g", G), ("d", D)):
        for n, b in mod.named_buffers():
            if n.endswith("weight_u") or n.endswith("weight_v"):
                put(f"sn/{tag}_buf/{n}", b)
    G.eval(), D.eval()
    with torch.no_grad():
        put("sn/eval/g_out_fp", fp(G(zc, v)))
        put("sn/eval/d_logit", D(mel_in, v))


def r1_cases():
    """_r1_penalty (lab3_train.py:329-339) on a training-mode discriminator with the projection head: value, the gradient of
    0.5 * gamma * penalty with respect to every parameter, BatchNorm buffers after the call; then the same through the multi-scale
    wrapper (value only), and two passes of train_stage1's D step with R1 on the first (loss_d, loss_r1, parameters after)."""
    D = RM.MelDiscriminator(base_channels=16, cond_dim=8)
    D.load_state_dict(fill_state_dict(D.state_dict(), 52))
    D.train()
    real = proc_input((4, 32, 64), "melr1.real", 0, 0.8)
    cond = proc_input((4, 8), "melr1.cond", 0)
    pen = RT._r1_penalty(D, real, cond)
    put("r1/penalty", pen)
    for (n, _), g in zip(D.named_parameters(), torch.autograd.grad(0.5 * 2.0 * pen, list(D.parameters()), allow_unused=True)):
        put(f"r1/grad/{n}", fp(g if g is not None else torch.zeros(1)))
    for n, b in D.named_buffers():
        if b.dtype.is_floating_point:
            put(f"r1/buf/{n}", b)
    MS = RM.MultiScaleMelDiscriminator(cond_dim=8, base_channels=16)
    MS.load_state_dict(fill_state_dict(MS.state_dict(), 53))
    put("r1/multiscale_penalty", RT._r1_penalty(MS.train(), proc_input((2, 64, 128), "melr1.real_ms", 0, 0.8), cond[:2]))
    # train_stage1 with R1 every second step, content / style terms off (no encoder needed): two batches
    G = RM.ReconstructionDecoder(**C.G_KW)
    G.load_state_dict(fill_state_dict(G.state_dict(), 21))
    D = RM.MelDiscriminator(base_channels=16, cond_dim=C.COND_DIM)
    D.load_state_dict(fill_state_dict(D.state_dict(), 22))
    cfg = RT.StageTrainConfig(stage_name="stage1", epochs=2, max_batches_per_epoch=1, gan_loss="hinge", r1_gamma=2.0, r1_interval=2,
                              weights=RT.TrainWeights(content=0.0, style=0.0), mrstft_resolutions=((16, 4, 16), (32, 8, 32), (64, 16, 64)))
    enc = C.ToyFrozenEncoder("cpu")
    cond_bank, _, _ = C.stage2_banks()
    with tempfile.TemporaryDirectory() as td:
        hist = RT.train_stage1(G, D, enc, C.ListLoader(C.stage2_batches(1), [0, 1, 2]), cond_bank, "cpu", cfg, Path(td) / "s1.pt")
    for ep in hist:
        for k, v in ep.items():
            if k != "stage":
                put(f"r1/stage1/hist/{ep['epoch']}/{k}", np.float64(v))
    for n, p_ in D.named_parameters():
        put(f"r1/stage1/d_param/{n}", fp(p_))
    print("r1", float(pen), {k: round(v, 5) for k, v in hist[0].items() if isinstance(v, float)})


def stage2_case(case: str):
    kw, _, _ = C.CASES[case]
    G = RM.ReconstructionDecoder(**C.G_KW)
    G.load_state_dict(fill_state_dict(G.state_dict(), 21))
    cond_d = C.COND_DIM + (C.ZC_DIM if kw.get("disc_use_content_cond") else 0)
    D = RM.MelDiscriminator(base_channels=16, cond_dim=cond_d)
    D.load_state_dict(fill_state_dict(D.state_dict(), 22))
    enc = C.ToyFrozenEncoder("cpu")
    cond_bank, proto, prof = C.stage2_banks()
    cfg = RT.StageTrainConfig(weights=RT.TrainWeights(**C.ALL_WEIGHTS), **kw)
    RT.StyleCritic = lambda **k: RM.StyleCritic(dropout=0.0, **k)
    with tempfile.TemporaryDirectory() as td:
        ck = Path(td) / "stage2.pt"
        torch.manual_seed(C.STAGE2_SEED)
        hist = RT.train_stage2(G, D, enc, C.loader_for(case), cond_bank, proto, None, prof, "cpu", cfg, ck, resume=False)
        payload = torch.load(str(ck), map_location="cpu", weights_only=True)  # written by the line above
    for ep in hist:
        for k, v in ep.items():
            if k != "stage":
                put(f"stage2/{case}/hist/{ep['epoch']}/{k}", np.float64(v))
    for group in ("generator", "discriminator", "style_critic", "frozen_encoder_model"):
        for k, v in payload[group].items():
            if v.dtype.is_floating_point:
                put(f"stage2/{case}/ckpt/{group}/{k}", fp(v))
    put(f"stage2/{case}/ckpt/keys", np.asarray(sorted(payload.keys())))
    print(case, {k: round(v, 5) for k, v in hist[-1].items() if isinstance(v, float)})


def main():
    penalty_cases()
    spectral_norm_cases()
    r1_cases()
    for case in C.CASES:
        stage2_case(case)
    path = HERE / "melstage2_golden.npz"
    np.savez_compressed(path, **STORE)
    print("wrote", path, len(STORE), "entries", path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
