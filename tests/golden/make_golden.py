#!/usr/bin/env python3
"""Generate golden vectors for the lab3 codec-track hot path FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference and the locally installed `transformers`):

    python tests/golden/make_golden.py

It imports the reference's own modules
  * /root/reference/lab 3/src/lab3_codec_models.py   (G, D, hinge / FM losses)
  * /root/reference/lab 3/src/lab3_codec_train.py    (_waveform_mrstft_loss; imported with empty
    stub modules for the absent `torchaudio` / `librosa`, inserted AFTER transformers resolved)
and the third-party SEANet decoder the reference calls through
`FrozenEncodec.decode_embeddings` (lab3_codec_bridge.py:135-146): transformers 5.15.0
`EncodecModel(EncodecConfig()).decoder` (random-init architecture; the pretrained weights are a remote
fetch and unavailable offline).  All parameters and inputs are filled procedurally
(oracle/procfill.py), so only small result vectors are committed - never reference source.

The training-step sequence is a hand-assembled loop over the imported reference pieces that follows
train_codec_stage (lab3_codec_train.py:359-604) line by line for the north-star loss subset
{adv, FM, latent L1, continuity, MR-STFT}; the content/style terms need the Lab1 encoder checkpoint
and torchaudio, neither of which exists here.  Random picks (noise, bank rows)
are injected as explicit procedural tensors.
"""
from __future__ import annotations

import json
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

from transformers import AutoFeatureExtractor, AutoModel, EncodecConfig, EncodecModel  # noqa: E402,F401

for _name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional", "librosa"):
    if _name not in sys.modules:
        sys.modules[_name] = types.ModuleType(_name)

from src import lab3_codec_models as RM  # noqa: E402
from src import lab3_codec_train as RT  # noqa: E402

from oracle.procfill import fill_state_dict, proc_input  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


# ----------------------------------------------------------------------------------------------
def build_models(z_style_dim=128, hidden=256, n_blocks=10, n_scales=3):
    G = RM.CodecLatentTranslator(z_style_dim=z_style_dim, hidden_channels=hidden, n_blocks=n_blocks)
    D = RM.MultiScaleWaveDiscriminator(n_scales)
    dec = EncodecModel(EncodecConfig()).decoder
    G.load_state_dict(fill_state_dict(G.state_dict(), 1))
    D.load_state_dict(fill_state_dict(D.state_dict(), 2))
    dec.load_state_dict(fill_state_dict(dec.state_dict(), 3))
    for p in dec.parameters():
        p.requires_grad = False
    dec.train(True)  # decode_embeddings flips this on around the call (lab3_codec_bridge.py:140-145)
    return G, D, dec


def make_inputs(B, T, tag, z_style_dim=128):
    q_src = proc_input((B, 128, T), f"{tag}.q_src", 0, 0.9)
    zc = F.normalize(proc_input((B, 128), f"{tag}.zc", 0), dim=-1)
    zs = F.normalize(proc_input((B, z_style_dim), f"{tag}.zs", 0), dim=-1)
    noise = proc_input((B, 32), f"{tag}.noise", 0, 1.7)
    q_real_d = proc_input((B, 128, T), f"{tag}.q_real_d", 0, 0.9)
    q_real_g = proc_input((B, 128, T), f"{tag}.q_real_g", 0, 0.9)
    return q_src, zc, zs, noise, q_real_d, q_real_g


def tsummary(t: torch.Tensor) -> np.ndarray:
    """[l2 norm, sum, abs-sum, first 4, last 4] - a compact fingerprint of a tensor."""
    f = t.detach().double().reshape(-1)
    head = f[:4].tolist() + [0.0] * (4 - min(4, f.numel()))
    tail = f[-4:].tolist() + [0.0] * (4 - min(4, f.numel()))
    return np.array([f.norm().item(), f.sum().item(), f.abs().sum().item()] + head[:4] + tail[:4], dtype=np.float64)


def params_summary(mod) -> dict:
    return {k: tsummary(p) for k, p in mod.named_parameters()}


def grads_summary(mod) -> dict:
    return {k: tsummary(p.grad) for k, p in mod.named_parameters() if p.grad is not None}


# ----------------------------------------------------------------------------------------------
def ref_iteration(G, D, dec, opt_g, opt_d, batch, global_step, cfg, record, grad_hook=None):
    """Mirror of lab3_codec_train.py:404-591 for the north-star loss subset.
    `grad_hook(kind, module)` (optional) runs right after each backward pass, before the clip."""
    q_src, zc_src, z_style_tgt, noise, q_real_tgt, q_real_tgt_g = batch
    q_hat = G(q_src=q_src, z_content=zc_src, z_style_tgt=z_style_tgt, noise=noise,
              style_dropout_p=0.0, style_jitter_std=0.0)
    x_hat = dec(q_hat)
    with torch.no_grad():
        x_src = dec(q_src)
    out = {}
    take_d_step = (global_step % max(1, int(cfg["d_step_period"]))) == 0
    out["take_d_step"] = float(take_d_step)
    if take_d_step:
        with torch.no_grad():
            x_real = dec(q_real_tgt)
        x_real = x_real.detach().requires_grad_(float(cfg["r1_gamma"]) > 0.0)
        real_outs = D(x_real)
        fake_outs_d = D(x_hat.detach())
        loss_d = RM.multiscale_hinge_d_loss(real_outs=real_outs, fake_outs=fake_outs_d)
        out["loss_d_hinge"] = float(loss_d.detach())
        do_r1 = float(cfg["r1_gamma"]) > 0.0 and (global_step % max(1, int(cfg["r1_interval"])) == 0)
        out["do_r1"] = float(do_r1)
        if do_r1:
            real_sum = torch.tensor(0.0)
            for real_logits, _ in real_outs:
                real_sum = real_sum + real_logits.mean()
            grad = torch.autograd.grad(outputs=real_sum, inputs=x_real, create_graph=True,
                                       retain_graph=True, only_inputs=True)[0]
            r1 = grad.reshape(grad.shape[0], -1).pow(2).sum(dim=1).mean()
            out["r1"] = float(r1.detach())
            loss_d = loss_d + 0.5 * float(cfg["r1_gamma"]) * r1
        opt_d.zero_grad(set_to_none=True)
        loss_d.backward()
        if record:
            out["d_grads"] = grads_summary(D)
        if grad_hook is not None:
            grad_hook("d", D)
        if float(cfg["d_grad_clip_norm"]) > 0.0:
            gn = torch.nn.utils.clip_grad_norm_(D.parameters(), float(cfg["d_grad_clip_norm"]))
            out["d_grad_norm"] = float(gn)
        opt_d.step()
        out["loss_d"] = float(loss_d.detach())

    with torch.no_grad():
        x_real_g = dec(q_real_tgt_g)
    real_outs_g = D(x_real_g)
    fake_outs = D(x_hat)
    loss_adv = RM.multiscale_hinge_g_loss(fake_outs=fake_outs)
    loss_fm = RM.multiscale_feature_matching_loss(real_outs=real_outs_g, fake_outs=fake_outs)
    loss_lat_l1 = F.l1_loss(q_hat, q_src)
    loss_lat_cont = F.l1_loss(q_hat[:, :, 1:] - q_hat[:, :, :-1], q_src[:, :, 1:] - q_src[:, :, :-1])
    loss_mrstft = RT._waveform_mrstft_loss(wav_pred=x_hat, wav_true=x_src,
                                           resolutions=cfg["resolutions"])
    delta_mean = (q_hat - q_src).abs().mean(dim=(1, 2))
    loss_delta_budget = F.relu(delta_mean - float(cfg["delta_budget"])).mean()
    w = cfg["weights"]
    loss_g = (float(w.adv) * loss_adv + float(w.feature_match) * loss_fm + float(w.latent_l1) * loss_lat_l1
              + float(w.latent_continuity) * loss_lat_cont + float(w.mrstft) * loss_mrstft
              + float(w.delta_budget) * loss_delta_budget)
    opt_g.zero_grad(set_to_none=True)
    loss_g.backward()
    if record:
        out["g_grads"] = grads_summary(G)
    if grad_hook is not None:
        grad_hook("g", G)
    if float(cfg["g_grad_clip_norm"]) > 0.0:
        gn = torch.nn.utils.clip_grad_norm_(G.parameters(), float(cfg["g_grad_clip_norm"]))
        out["g_grad_norm"] = float(gn)
    opt_g.step()
    out.update(loss_g=float(loss_g.detach()), loss_adv=float(loss_adv.detach()), loss_fm=float(loss_fm.detach()),
               loss_lat_l1=float(loss_lat_l1.detach()), loss_lat_cont=float(loss_lat_cont.detach()),
               loss_mrstft=float(loss_mrstft.detach()), loss_delta_budget=float(loss_delta_budget.detach()))
    if record:
        out["q_hat"] = q_hat.detach().numpy().copy()
        out["x_hat"] = x_hat.detach().numpy().copy()
        out["x_src"] = x_src.detach().numpy().copy()
    return out


def flatten(prefix, d, store):
    for k, v in d.items():
        key = f"{prefix}/{k}"
        if isinstance(v, dict):
            flatten(key, v, store)
        else:
            store[key] = np.asarray(v)


# ----------------------------------------------------------------------------------------------
def case_forward_small(store):
    """B=2, T=16 (S=5120): module forward outputs + first-order gradients of every loss term."""
    G, D, dec = build_models()
    q_src, zc, zs, noise, q_real_d, q_real_g = make_inputs(2, 16, "small")
    q_hat = G(q_src, zc, zs, noise)
    store["fwd/q_hat"] = q_hat.detach().numpy()
    # generator options: dilations for n_blocks != 10, direct_output, noise=None
    x_hat = dec(q_hat)
    store["fwd/x_hat"] = x_hat.detach().numpy()
    # per-layer decoder fingerprints (HF layer order)
    h = q_hat.detach()
    for i, layer in enumerate(dec.layers):
        h = layer(h)
        store[f"fwd/dec_layer{i}"] = tsummary(h)
    outs = D(x_hat)
    for s, (lg, feats) in enumerate(outs):
        store[f"fwd/d_logits{s}"] = lg.detach().numpy()
        for j, f in enumerate(feats):
            store[f"fwd/d_feat{s}_{j}"] = tsummary(f)
            store[f"fwd/d_feat{s}_{j}_slice"] = f.detach()[:, ::7, ::13].numpy()
    with torch.no_grad():
        x_real = dec(q_real_d)
    outs_r = D(x_real)
    store["loss/hinge_d"] = np.float64(RM.multiscale_hinge_d_loss(outs_r, outs).item())
    store["loss/hinge_g"] = np.float64(RM.multiscale_hinge_g_loss(outs).item())
    store["loss/fm"] = np.float64(RM.multiscale_feature_matching_loss(outs_r, outs).item())
    store["loss/hinge_d_single"] = np.float64(RM.hinge_d_loss(outs_r[0][0], outs[0][0]).item())
    store["loss/hinge_g_single"] = np.float64(RM.hinge_g_loss(outs[0][0]).item())
    store["loss/fm_single"] = np.float64(RM.feature_matching_loss(outs_r[0][1], outs[0][1]).item())
    # gradient of each G-side loss w.r.t. x_hat and q_hat, separately
    for name, loss in (
        ("hinge_g", RM.multiscale_hinge_g_loss(outs)),
        ("fm", RM.multiscale_feature_matching_loss(outs_r, outs)),
    ):
        gx, = torch.autograd.grad(loss, x_hat, retain_graph=True)
        store[f"grad/{name}_wrt_x_hat"] = gx.numpy()
    with torch.no_grad():
        x_src = dec(q_src)
    l_st = RT._waveform_mrstft_loss(x_hat, x_src, ((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048)))
    store["loss/mrstft"] = np.float64(l_st.item())
    gx, = torch.autograd.grad(l_st, x_hat, retain_graph=True)
    store["grad/mrstft_wrt_x_hat"] = gx.numpy()
    gq, = torch.autograd.grad(x_hat.square().mean(), q_hat, retain_graph=True)
    store["grad/dec_sqmean_wrt_q_hat"] = gq.numpy()
    # generator variants
    G2 = RM.CodecLatentTranslator(z_style_dim=160, hidden_channels=192, n_blocks=8, direct_output=True)
    G2.load_state_dict(fill_state_dict(G2.state_dict(), 11))
    zs160 = F.normalize(proc_input((2, 160), "small.zs160", 0), dim=-1)
    store["fwd/q_hat_variant"] = G2(q_src, zc, zs160, None).detach().numpy()


def case_step_sequence(store, meta):
    """4 consecutive iterations, d_step_period=2, r1_interval=2, r1_gamma=1, clip 5.0, AdamW.
    The MR-STFT weight is 0 here on purpose: the gradient of its log-magnitude term is ill-conditioned
    (1e-6 input noise moves it by ~0.4 % even in fp64, because low-magnitude bins dominate it and flip sign),
    which makes multi-iteration trajectories chaotic at the 1e-4 level.  That term's value and gradient are
    pinned at the reference's own x_hat in case_forward_small / case_step_period1 instead."""
    G, D, dec = build_models()
    G.train(); D.train()
    cfg = dict(d_step_period=2, r1_gamma=1.0, r1_interval=2, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
               delta_budget=0.12, resolutions=((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048)),
               weights=RM.CodecTrainWeights(adv=0.55, latent_l1=0.2, latent_continuity=1.0, content=0.0, style=0.0,
                                            mrstft=0.0, feature_match=1.0, delta_budget=1.0))
    meta["step_sequence_cfg"] = {k: (v if not hasattr(v, "__dict__") else v.__dict__) for k, v in cfg.items()}
    opt_g = torch.optim.AdamW(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt_d = torch.optim.AdamW(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    for it in range(1, 5):
        batch = make_inputs(2, 16, f"seq{it}")
        out = ref_iteration(G, D, dec, opt_g, opt_d, batch, global_step=it, cfg=cfg, record=True)
        flatten(f"seq/it{it}", out, store)
        flatten(f"seq/it{it}/g_params", params_summary(G), store)
        flatten(f"seq/it{it}/d_params", params_summary(D), store)


def case_step_period1(store):
    """One iteration of the headline unit of work: d_step_period=1, no R1, default stage weights."""
    G, D, dec = build_models()
    G.train(); D.train()
    cfg = dict(d_step_period=1, r1_gamma=0.0, r1_interval=16, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
               delta_budget=0.12, resolutions=((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048)),
               weights=RM.CodecTrainWeights(content=0.0, style=0.0))
    opt_g = torch.optim.AdamW(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt_d = torch.optim.AdamW(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    batch = make_inputs(2, 16, "p1")
    out = ref_iteration(G, D, dec, opt_g, opt_d, batch, global_step=1, cfg=cfg, record=True)
    flatten("p1", out, store)
    flatten("p1/g_params", params_summary(G), store)
    flatten("p1/d_params", params_summary(D), store)


def case_d_only(store):
    """D-only at S=16384 (odd feature lengths through the stride-4 stack), incl. input gradient."""
    _, D, _ = build_models()
    wav = proc_input((2, 1, 16384), "donly.wav", 0, 0.8).requires_grad_(True)
    outs = D(wav)
    tot = torch.tensor(0.0)
    for s, (lg, feats) in enumerate(outs):
        store[f"donly/logits{s}"] = lg.detach().numpy()
        for j, f in enumerate(feats):
            store[f"donly/feat{s}_{j}"] = tsummary(f)
        tot = tot + lg.mean() + sum(f.square().mean() for f in feats)
    tot.backward()
    store["donly/grad_wav"] = wav.grad[:, :, ::17].numpy()
    store["donly/grad_wav_sum"] = tsummary(wav.grad)
    flatten("donly/grads", grads_summary(D), store)
    # odd length, not a multiple of 4 anywhere
    wav2 = proc_input((1, 1, 3001), "donly.wav2", 0, 0.8)
    outs2 = D(wav2)
    for s, (lg, feats) in enumerate(outs2):
        store[f"donly_odd/logits{s}"] = lg.detach().numpy()
        store[f"donly_odd/feat_shapes{s}"] = np.array([list(f.shape) for f in feats], dtype=np.int64)


def case_mrstft(store):
    """Stand-alone MR-STFT: full-length signals, skipped resolutions, L1 fallback, sibling formula."""
    x = proc_input((2, 1, 120000), "stft.x", 0, 0.5)
    y = proc_input((2, 1, 120000), "stft.y", 0, 0.5)
    res = ((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048))
    store["mrstft/full"] = np.float64(RT._waveform_mrstft_loss(x, y, res).item())
    xs = x[:, :, :1500].clone().requires_grad_(True)
    l = RT._waveform_mrstft_loss(xs, y[:, :, :1500], res)  # 2048 skipped
    store["mrstft/short1500"] = np.float64(l.item())
    l.backward()
    store["mrstft/short1500_grad"] = xs.grad.numpy()
    store["mrstft/fallback_l1"] = np.float64(RT._waveform_mrstft_loss(x[:, :, :300], y[:, :, :300], res).item())
    # near-silent input exercises clamp_min(1e-8)
    z = torch.zeros(1, 1, 4096)
    z[0, 0, 100] = 1e-3
    store["mrstft/silence"] = np.float64(RT._waveform_mrstft_loss(z, x[:1, :, :4096], res).item())
    # long-clip stress resolutions (config 5), short signal so the fixture stays cheap
    res5 = ((256, 64, 256), (512, 128, 512), (1024, 256, 1024), (2048, 512, 2048), (4096, 1024, 4096), (8192, 2048, 8192))
    store["mrstft/stress"] = np.float64(RT._waveform_mrstft_loss(x[:, :, :30000], y[:, :, :30000], res5).item())
    # per-resolution sc / lm on a short pair (restated here from the reference lines 165-171)
    for n_fft, hop, win in res:
        w = torch.hann_window(win)
        sx = torch.stft(x[:, 0, :8192], n_fft=n_fft, hop_length=hop, win_length=win, window=w, return_complex=True, center=True)
        sy = torch.stft(y[:, 0, :8192], n_fft=n_fft, hop_length=hop, win_length=win, window=w, return_complex=True, center=True)
        mx = sx.abs().clamp_min(1e-8); my = sy.abs().clamp_min(1e-8)
        store[f"mrstft/sc_{n_fft}"] = np.float64((torch.norm(my - mx, p="fro") / (torch.norm(my, p="fro") + 1e-8)).item())
        store[f"mrstft/lm_{n_fft}"] = np.float64(F.l1_loss(torch.log(mx), torch.log(my)).item())
        store[f"mrstft/mag_{n_fft}"] = tsummary(mx)


def case_full_shape(store):
    """Full codec shape (T=375, S=120000), B=1: scalars and fingerprints only."""
    G, D, dec = build_models()
    q_src, zc, zs, noise, q_real_d, _ = make_inputs(1, 375, "full")
    with torch.no_grad():
        q_hat = G(q_src, zc, zs, noise)
        x_hat = dec(q_hat)
        x_real = dec(q_real_d)
        outs = D(x_hat)
        outs_r = D(x_real)
        store["full/q_hat"] = tsummary(q_hat)
        store["full/x_hat"] = tsummary(x_hat)
        store["full/x_hat_slice"] = x_hat[:, :, ::997].numpy()
        for s, (lg, feats) in enumerate(outs):
            store[f"full/d_logits{s}"] = lg.numpy()
            for j, f in enumerate(feats):
                store[f"full/d_feat{s}_{j}"] = tsummary(f)
        store["full/hinge_d"] = np.float64(RM.multiscale_hinge_d_loss(outs_r, outs).item())
        store["full/fm"] = np.float64(RM.multiscale_feature_matching_loss(outs_r, outs).item())
        store["full/mrstft"] = np.float64(RT._waveform_mrstft_loss(x_hat, x_real, RT.CodecStageTrainConfig("s", 1).wave_mrstft_resolutions).item())


def case_banks(store):
    """Bank builders + target sampling helpers (host logic on the path)."""
    zs = proc_input((10, 128), "bank.zs", 0).numpy()
    gi = np.array([0, 1, 2, 3, 0, 1, 2, 0, 1, 0], dtype=np.int64)
    store["bank/centroid"] = RT.build_style_centroid_bank(zs, gi, 5).numpy()
    ex = RT.build_style_exemplar_bank(zs, gi, 5)
    store["bank/exemplar_keys"] = np.array(sorted(ex.keys()), dtype=np.int64)
    store["bank/exemplar0"] = ex[0].numpy()
    src = torch.tensor([0, 1, 2, 3, 3, 2, 1, 0])
    store["bank/shift_balanced"] = RT._sample_shift_targets_balanced(src, 4, offset=3).numpy()
    cent = torch.from_numpy(store["bank/centroid"])
    z = RT._mix_style_condition(torch.tensor([0, 1, 2]), "mix", 0.7, cent, None, torch.device("cpu"), 0.0)
    store["bank/mix_centroid_only"] = z.numpy()


def main():
    store, meta = {}, {}
    case_forward_small(store)
    case_step_period1(store)
    case_step_sequence(store, meta)
    case_d_only(store)
    case_mrstft(store)
    case_full_shape(store)
    case_banks(store)
    out = HERE / "codec_golden.npz"
    np.savez_compressed(out, **store)
    meta.update(torch=torch.__version__, transformers=__import__("transformers").__version__,
                n_entries=len(store), generator="tests/golden/make_golden.py")
    (HERE / "codec_golden_meta.json").write_text(json.dumps(meta, indent=1, default=str))
    print(f"wrote {out} ({out.stat().st_size/1e6:.2f} MB, {len(store)} entries)")


if __name__ == "__main__":
    main()
