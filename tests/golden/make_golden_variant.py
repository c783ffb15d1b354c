#!/usr/bin/env python3
"""Golden vectors for BASELINE config 4 (genre-conditioned variant), FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference and transformers):

    python tests/golden/make_golden_variant.py        # writes tests/golden/variant_golden.npz

The CLAP-recipe model of `scripts/run_codec_clap_labels_full.ps1:12-15` with the lab2 target160 conditioning
(`lab 2/src/lab2_pipeline.py:80-93`): CodecLatentTranslator(z_style_dim=160, hidden_channels=192, n_blocks=8),
MultiScaleWaveDiscriminator(2), B=2, with the stage-3 defaults style_dropout_p=0.25 / style_jitter_std=0.08
(`run_lab3_codec.py:264-265`).  The reference draws the dropout mask and the jitter INSIDE forward
(lab3_codec_models.py:104-108: torch.rand((B,1)) then torch.randn_like(z_style)); this script seeds the CPU generator,
records those two draws by making them first, re-seeds, and runs the reference forward, so the fixtures hold the mask and
the jitter the output was computed with (`v4/style_keep`, `v4/style_jitter`).

One G step on that model (adv + FM + latent L1 + continuity, D frozen) and one D step, with gradient fingerprints of every
parameter: the imported reference pieces assembled as train_codec_stage does (lab3_codec_train.py:404-470,573-591).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import make_golden as MG  # noqa: E402  (imports the reference modules with the torchaudio / librosa stubs)
from make_golden import RM, fill_state_dict, flatten, grads_summary, proc_input, tsummary  # noqa: E402
from transformers import EncodecConfig, EncodecModel  # noqa: E402

P_DROP, JITTER = 0.25, 0.08
SEED = 1234


def main():
    store = {}
    G = RM.CodecLatentTranslator(z_style_dim=160, hidden_channels=192, n_blocks=8)
    D = RM.MultiScaleWaveDiscriminator(2)
    dec = EncodecModel(EncodecConfig()).decoder
    G.load_state_dict(fill_state_dict(G.state_dict(), 21))
    D.load_state_dict(fill_state_dict(D.state_dict(), 22))
    dec.load_state_dict(fill_state_dict(dec.state_dict(), 3))
    for p in dec.parameters():
        p.requires_grad = False
    dec.train(True)
    G.train(); D.train()
    B, T = 2, 16
    q_src, zc, zs160, noise, q_real_d, q_real_g = MG.make_inputs(B, T, "v4", z_style_dim=160)

    # the two draws forward() will make, in its order; one of the two samples is forced through the dropout branch by the seed
    torch.manual_seed(SEED)
    keep = (torch.rand((B, 1)) >= P_DROP).float()
    jitter = torch.randn_like(zs160) * JITTER
    store["v4/style_keep"] = keep.numpy()
    store["v4/style_jitter"] = jitter.numpy()
    torch.manual_seed(SEED)
    q_hat = G(q_src=q_src, z_content=zc, z_style_tgt=zs160, noise=noise, style_dropout_p=P_DROP, style_jitter_std=JITTER)
    store["v4/q_hat"] = q_hat.detach().numpy()
    x_hat = dec(q_hat)
    store["v4/x_hat"] = x_hat.detach().numpy()
    with torch.no_grad():
        x_real = dec(q_real_d)
        x_real_g = dec(q_real_g)

    # D step (hinge) on the 2-scale discriminator
    real_outs, fake_outs_d = D(x_real), D(x_hat.detach())
    loss_d = RM.multiscale_hinge_d_loss(real_outs=real_outs, fake_outs=fake_outs_d)
    store["v4/loss_d"] = np.float64(loss_d.item())
    for s, (lg, feats) in enumerate(fake_outs_d):
        store[f"v4/d_logits{s}"] = lg.detach().numpy()
        for j, f in enumerate(feats):
            store[f"v4/d_feat{s}_{j}"] = tsummary(f)
    D.zero_grad(set_to_none=True)
    loss_d.backward()
    flatten("v4/d_grads", grads_summary(D), store)

    # G step terms (D's parameters receive discarded gradients in the reference; only G's are recorded)
    real_outs_g, fake_outs = D(x_real_g), D(x_hat)
    loss_adv = RM.multiscale_hinge_g_loss(fake_outs=fake_outs)
    loss_fm = RM.multiscale_feature_matching_loss(real_outs=real_outs_g, fake_outs=fake_outs)
    loss_l1 = F.l1_loss(q_hat, q_src)
    loss_cont = F.l1_loss(q_hat[:, :, 1:] - q_hat[:, :, :-1], q_src[:, :, 1:] - q_src[:, :, :-1])
    w = RM.CodecTrainWeights()
    loss_g = float(w.adv) * loss_adv + float(w.feature_match) * loss_fm + float(w.latent_l1) * loss_l1 \
        + float(w.latent_continuity) * loss_cont
    for k, v in (("loss_adv", loss_adv), ("loss_fm", loss_fm), ("loss_lat_l1", loss_l1), ("loss_lat_cont", loss_cont), ("loss_g", loss_g)):
        store[f"v4/{k}"] = np.float64(v.item())
    G.zero_grad(set_to_none=True)
    loss_g.backward()
    flatten("v4/g_grads", grads_summary(G), store)
    store["v4/g_grad_flat_norm"] = np.float64(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in G.parameters())).item())

    out = HERE / "variant_golden.npz"
    np.savez_compressed(out, **store)
    print(f"wrote {out} ({out.stat().st_size / 1e3:.1f} kB, {len(store)} entries); keep = {keep.flatten().tolist()}")


if __name__ == "__main__":
    main()
