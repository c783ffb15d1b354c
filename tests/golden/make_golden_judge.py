"""Golden vectors for the codec style judge term (style half): imports the REFERENCE's own
`lab 3/src/lab3_codec_judge.py::CodecStyleJudge`, fills it procedurally (oracle/procfill.py), and records its embedding,
logits and the style terms of the G step exactly as the reference computes them (lab3_codec_train.py:480-494), plus their
gradient with respect to q_hat.  Run in the build container only (the reference does not travel):
    python tests/golden/make_golden_judge.py   ->  tests/golden/judge_golden.npz
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/lab 3")

from src import lab3_codec_judge as RJ  # noqa: E402

from oracle.procfill import fill_state_dict, proc_input  # noqa: E402


def main():
    torch.manual_seed(0)
    store = {}
    for tag, (B, T, n_genres, hidden, emb) in {"small": (3, 37, 5, 64, 128), "full": (2, 375, 10, 256, 128)}.items():
        judge = RJ.CodecStyleJudge(in_channels=128, n_genres=n_genres, hidden=hidden, emb_dim=emb)
        judge.load_state_dict(fill_state_dict(judge.state_dict(), 11))
        judge.eval()
        q_hat = proc_input((B, 128, T), f"judge.{tag}.q", 0, 0.9).requires_grad_(True)
        z_style_tgt = proc_input((B, emb), f"judge.{tag}.zs", 0)
        tgt = torch.tensor([(2 * i + 1) % n_genres for i in range(B)])
        src = torch.tensor([(i + 3) % n_genres for i in range(B)])
        align_w, margin = 0.5, 0.30
        # lab3_codec_train.py:480-494, mode codec_judge_ce, stage2
        emb_hat = judge.embed(q_hat)
        logits = judge.head(emb_hat)
        loss_style = F.cross_entropy(logits, tgt)
        z_ref = F.normalize(z_style_tgt, dim=-1)
        loss_style_embed = (1.0 - F.cosine_similarity(emb_hat, z_ref, dim=-1)).mean()
        loss_style = loss_style + align_w * loss_style_embed
        probs = torch.softmax(logits, dim=1)
        p_src = probs.gather(1, src.view(-1, 1)).squeeze(1)
        loss_style_push = F.relu(p_src - margin).mean()
        total = 2.0 * loss_style + 0.7 * loss_style_push
        g, = torch.autograd.grad(total, q_hat)
        store[f"{tag}/emb"] = emb_hat.detach().numpy()
        store[f"{tag}/logits"] = logits.detach().numpy()
        store[f"{tag}/loss_style"] = np.float64(loss_style.item())
        store[f"{tag}/loss_style_embed"] = np.float64(loss_style_embed.item())
        store[f"{tag}/loss_style_push"] = np.float64(loss_style_push.item())
        store[f"{tag}/grad_q"] = g.numpy()
        store[f"{tag}/cfg"] = np.array([B, T, n_genres, hidden, emb], dtype=np.int64)
        store[f"{tag}/tgt"] = tgt.numpy()
        store[f"{tag}/src"] = src.numpy()
        store[f"{tag}/state_keys"] = np.array(sorted(judge.state_dict().keys()))
    np.savez_compressed(REPO / "tests" / "golden" / "judge_golden.npz", **store)
    print("wrote", len(store), "entries")


if __name__ == "__main__":
    main()
