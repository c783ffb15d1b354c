#!/usr/bin/env python3
"""Golden vectors for the Lab1 `ChunkEncoder` (the frozen encoder of the G step's content / style terms), FROM THE REFERENCE.

Run in the build container only:   python tests/golden/make_golden_lab1.py      # writes tests/golden/lab1_golden.npz

Imports the reference's `lab 3/src/lab3_bridge.py` (ChunkEncoder, :34-76; `librosa` is absent and replaced by an empty stub
module, nothing of it is called), fills it procedurally (oracle.lab1_oracle.fill_chunk_encoder), puts it in eval mode as
FrozenLab1Encoder does (:186-190), and records `forward_log_mel_tensor`'s z_content / z_style for a procedural log-mel input
in dB range, the content cosine loss of lab3_codec_train.py:474-475 against a procedural target, and its gradient w.r.t. the
log-mel input.  The waveform -> log-mel adapter in front of it is torchaudio code and cannot be generated here (parity
unpinned, see oracle/lab1_oracle.py).
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
if "librosa" not in sys.modules:
    sys.modules["librosa"] = types.ModuleType("librosa")

from src import lab3_bridge as RB  # noqa: E402

from oracle.lab1_oracle import fill_chunk_encoder  # noqa: E402
from oracle.procfill import proc_input  # noqa: E402


def main():
    enc = RB.ChunkEncoder(n_sources=5, z_dim=128)
    enc.load_state_dict(fill_chunk_encoder(enc.state_dict(), 31))
    enc.eval()
    for p in enc.parameters():
        p.requires_grad = False
    log_mel = (proc_input((3, 96, 256), "lab1.logmel", 0, 40.0) - 40.0).requires_grad_(True)   # [-80, 0) dB
    zc_src = F.normalize(proc_input((3, 128), "lab1.zc_src", 0), dim=-1)
    out = enc(log_mel, grl_lambda=0.0)
    zc_hat = F.normalize(out["z_content"], dim=-1)
    loss = (1.0 - F.cosine_similarity(zc_hat, zc_src, dim=-1)).mean()
    g, = torch.autograd.grad(loss, log_mel)
    store = {"z_content": out["z_content"].detach().numpy(), "z_style": out["z_style"].detach().numpy(),
             "shared_feat": out["shared_feat"].detach().numpy(), "loss_content": np.float64(loss.item()), "grad_log_mel": g.numpy()[:, ::5, ::7]}
    path = HERE / "lab1_golden.npz"
    np.savez_compressed(path, **store)
    print("wrote", path, path.stat().st_size, "bytes; loss_content", loss.item())


if __name__ == "__main__":
    main()
