"""GPU: gan_amd.lab3_bridge.ChunkEncoder / FrozenLab1Encoder (the frozen Lab1 encoder with its three convolutions on the HIP kernels)
against the vectors recorded from the reference's own ChunkEncoder (tests/golden/lab1_golden.npz, made by make_golden_lab1.py from
lab 3/src/lab3_bridge.py:34-76 in eval mode): latents, shared features, the content cosine loss of lab3_codec_train.py:474-475 and its
gradient with respect to the log-mel input.  fp32 mode rtol 1e-4; bf16 mode within the stated bound (latent cosine >= 0.999)."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_amd import lab3_bridge as PB
from oracle.lab1_oracle import fill_chunk_encoder
from oracle.procfill import proc_input

DEV = torch.device("cuda")
GOLD = np.load(Path(__file__).parent / "golden" / "lab1_golden.npz")


def filled(precision):
    enc = PB.ChunkEncoder(n_sources=5, z_dim=128, precision=precision)
    enc.load_state_dict(fill_chunk_encoder(enc.state_dict(), 31))
    enc = enc.to(DEV).eval()
    for p in enc.parameters():
        p.requires_grad = False
    return enc


def test_chunk_encoder_matches_the_reference_in_fp32():
    enc = filled("fp32")
    log_mel = (proc_input((3, 96, 256), "lab1.logmel", 0, 40.0) - 40.0).to(DEV).requires_grad_(True)
    zc_src = F.normalize(proc_input((3, 128), "lab1.zc_src", 0), dim=-1).to(DEV)
    out = enc(log_mel, grl_lambda=0.0)
    assert set(out) == {"shared_feat", "z_content", "z_style", "style_logits", "content_style_logits", "music_logit"}
    for k in ("z_content", "z_style", "shared_feat"):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), GOLD[k], rtol=1e-4, atol=2e-6, err_msg=k)
    loss = (1.0 - F.cosine_similarity(F.normalize(out["z_content"], dim=-1), zc_src, dim=-1)).mean()
    np.testing.assert_allclose(float(loss.detach()), float(GOLD["loss_content"]), rtol=1e-5)
    g, = torch.autograd.grad(loss, log_mel)
    want = GOLD["grad_log_mel"]
    got = g.cpu().numpy()[:, ::5, ::7]
    # ReLU kinks: an entry whose pre-activation is ~0 may take the other branch; bound the slice in relative L2 and by outlier count
    assert np.linalg.norm(got - want) <= 1e-3 * np.linalg.norm(want)
    assert (np.abs(got - want) > 1e-4 * np.abs(want).max()).mean() <= 0.01


def test_chunk_encoder_bf16_bound():
    enc32, enc16 = filled("fp32"), filled("bf16")
    log_mel = (proc_input((8, 96, 256), "lab1.logmel8", 0, 40.0) - 40.0).to(DEV)
    with torch.no_grad():
        a, b = enc32(log_mel, grl_lambda=0.0), enc16(log_mel, grl_lambda=0.0)
    for k in ("z_content", "z_style"):
        cos = F.cosine_similarity(a[k], b[k], dim=-1)
        assert float(cos.min()) >= 0.999, (k, float(cos.min()))
    assert float((a["shared_feat"] - b["shared_feat"]).abs().max()) <= 3e-2 * float(a["shared_feat"].abs().max())


def test_frozen_lab1_encoder_loads_a_checkpoint_and_serves_both_tracks(tmp_path):
    sd = fill_chunk_encoder(PB.ChunkEncoder(n_sources=3, z_dim=128).state_dict(), 31)
    ck = tmp_path / "lab1.pt"
    torch.save({"cfg": {"sample_rate": 22050, "chunk_seconds": 5.0, "z_dim": 128}, "source_to_idx": {"a": 0, "b": 1, "c": 2}, "model": sd}, ck)
    enc = PB.FrozenLab1Encoder(ck, device="cuda", precision="bf16")
    assert enc.cfg.sample_rate == 22050 and enc.cfg.z_dim == 128 and not enc.model.training
    assert all(not p.requires_grad for p in enc.model.parameters())
    mel = (proc_input((2, 96, 256), "lab1.logmel2", 0, 40.0) - 40.0)
    lat = enc.infer_log_mel(mel[0].numpy())
    assert lat["z_content"].shape == (128,) and lat["z_content"].dtype == np.float32 and abs(np.linalg.norm(lat["z_style"]) - 1.0) < 1e-3
    x = mel.to(DEV).requires_grad_(True)
    out = enc.forward_log_mel_tensor(x)
    np.testing.assert_allclose(out["z_content"][0].detach().cpu().numpy(), lat["z_content"], atol=2e-3)
    out["z_style"].square().sum().backward()  # ||z|| = 1: a constant; use a non-trivial functional below
    (enc.forward_log_mel_tensor(x)["shared_feat"].sum()).backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    with pytest.raises(FileNotFoundError):
        PB.FrozenLab1Encoder(tmp_path / "absent.pt")
    torch.save({"cfg": {}, "model": sd}, tmp_path / "bad.pt")
    with pytest.raises(ValueError):
        PB.FrozenLab1Encoder(tmp_path / "bad.pt")
    with pytest.raises(NotImplementedError):
        enc.infer_file_with_mel("x.wav", 256)
