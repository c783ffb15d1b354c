#!/usr/bin/env python3
"""Golden GRADIENT vectors at the full codec shape (T=375, S=120000), FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference and transformers):

    python tests/golden/make_golden_fullgrad.py       # writes tests/golden/fullgrad_golden.npz  (~1 min of CPU)

One iteration of the headline unit of work (d_step_period=1, no R1, clip 5.0, AdamW) at B=2 on the full-size models,
assembled from the imported reference pieces exactly as make_golden.ref_iteration does (lab3_codec_train.py:404-591):

  fullgrad/loss_*            every loss term of the D step and the G step
  fullgrad/{d,g}_grad_norm   the clip norms
  fullgrad/{d,g}_grads/<p>   fingerprint [l2, sum, abs-sum, first 4, last 4] of every parameter's gradient (before the clip)
  fullgrad/{d,g}_grad_sub    every STRIDE-th element of the flat gradient vector (named_parameters order, before the
                             clip): what the bf16 cosine / relative-L2 bound is evaluated on (the whole vectors are
                             86 MB + 49 MB; strides 499 / 251 are coprime to every kernel size and channel count)
  fullgrad/{d,g}_params/<p>  fingerprint of every parameter after its AdamW step
  fullgrad/x_hat, q_hat      fingerprints of the forward results the gradients were taken at

The MR-STFT WEIGHT is 0 (its value is still recorded): that term's gradient is ill-conditioned in any precision
(tests/test_oracle_golden.py::test_mrstft_gradient_is_ill_conditioned_in_fp32), so a fixture of it would pin noise.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import make_golden as MG  # noqa: E402  (imports the reference modules with the torchaudio / librosa stubs)
from make_golden import RM, flatten, params_summary, tsummary  # noqa: E402

STRIDE = {"d": 499, "g": 251}


def flat_sub(mod, stride):
    flat = torch.cat([p.grad.detach().reshape(-1) for _, p in mod.named_parameters()])
    return flat[::stride].clone().numpy()


def main():
    store = {}
    G, D, dec = MG.build_models()
    G.train(); D.train()
    cfg = dict(d_step_period=1, r1_gamma=0.0, r1_interval=16, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
               delta_budget=0.12, resolutions=((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048)),
               weights=RM.CodecTrainWeights(content=0.0, style=0.0, mrstft=0.0))
    opt_g = torch.optim.AdamW(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt_d = torch.optim.AdamW(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    batch = MG.make_inputs(2, 375, "fullgrad")

    def hook(kind, mod):
        store[f"fullgrad/{kind}_grad_sub"] = flat_sub(mod, STRIDE[kind])
        store[f"fullgrad/{kind}_grad_flat_norm"] = np.float64(
            torch.sqrt(sum(p.grad.double().pow(2).sum() for p in mod.parameters())).item())

    out = MG.ref_iteration(G, D, dec, opt_g, opt_d, batch, global_step=1, cfg=cfg, record=True, grad_hook=hook)
    for k in ("q_hat", "x_hat", "x_src"):
        store[f"fullgrad/{k}"] = tsummary(torch.from_numpy(out.pop(k)))
    flatten("fullgrad", out, store)
    flatten("fullgrad/g_params", params_summary(G), store)
    flatten("fullgrad/d_params", params_summary(D), store)
    store["fullgrad/stride_d"] = np.int64(STRIDE["d"])
    store["fullgrad/stride_g"] = np.int64(STRIDE["g"])
    dst = HERE / "fullgrad_golden.npz"
    np.savez_compressed(dst, **store)
    print(f"wrote {dst} ({dst.stat().st_size / 1e3:.1f} kB, {len(store)} entries)")
    for k in ("loss_d", "loss_g", "loss_adv", "loss_fm", "loss_mrstft", "d_grad_norm", "g_grad_norm"):
        print(k, out[k])


if __name__ == "__main__":
    main()
