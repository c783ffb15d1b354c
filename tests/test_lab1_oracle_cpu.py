"""CPU: the Lab1 content-path oracle (oracle/lab1_oracle.py).  The ChunkEncoder restatement is pinned by vectors the reference's
own class produced (tests/golden/lab1_golden.npz); the torchaudio-shaped waveform -> log-mel adapter is PARITY UNPINNED (torchaudio
is not importable here and the reference holds no fixture), so it is checked against independent mathematics instead: an exactly
resampled sinusoid, torch.stft, and the filterbank's defining properties."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lab1_oracle as LO
from oracle.procfill import proc_input

GOLD = Path(__file__).resolve().parent / "golden"


def _encoder_sd():
    shapes = {"backbone.0.weight": (32, 1, 3, 3), "backbone.0.bias": (32,), "backbone.3.weight": (64, 32, 3, 3), "backbone.3.bias": (64,),
              "backbone.6.weight": (128, 64, 3, 3), "backbone.6.bias": (128,), "shared.weight": (256, 128), "shared.bias": (256,),
              "content_head.weight": (128, 256), "content_head.bias": (128,), "style_head.weight": (128, 256), "style_head.bias": (128,),
              "style_cls.weight": (5, 128), "style_cls.bias": (5,), "content_style_adv.0.weight": (128, 128), "content_style_adv.0.bias": (128,),
              "content_style_adv.2.weight": (5, 128), "content_style_adv.2.bias": (5,), "music_head.weight": (1, 256), "music_head.bias": (1,)}
    for i, c in ((1, 32), (4, 64), (7, 128)):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"backbone.{i}.{leaf}"] = (c,)
    return LO.fill_chunk_encoder({k: torch.empty(s) for k, s in shapes.items()}, 31)


def test_chunk_encoder_matches_reference_golden():
    with np.load(GOLD / "lab1_golden.npz") as z:
        gold = {k: z[k] for k in z.files}
    sd = _encoder_sd()
    log_mel = (proc_input((3, 96, 256), "lab1.logmel", 0, 40.0) - 40.0).requires_grad_(True)
    zc_src = F.normalize(proc_input((3, 128), "lab1.zc_src", 0), dim=-1)
    out = LO.chunk_encoder_forward(sd, log_mel)
    for k in ("z_content", "z_style", "shared_feat"):
        np.testing.assert_allclose(out[k].detach().numpy(), gold[k], rtol=1e-4, atol=1e-6, err_msg=k)
    loss = (1.0 - F.cosine_similarity(F.normalize(out["z_content"], dim=-1), zc_src, dim=-1)).mean()
    assert float(loss) == pytest.approx(float(gold["loss_content"]), rel=1e-5)
    g, = torch.autograd.grad(loss, log_mel)
    np.testing.assert_allclose(g.numpy()[:, ::5, ::7], gold["grad_log_mel"], rtol=1e-3, atol=1e-9)


def test_resample_of_a_sinusoid_is_that_sinusoid_at_the_new_rate():
    """Windowed-sinc resampling (24 kHz -> 22.05 kHz = 160 -> 147) of a band-limited tone: away from the edges the output is the
    same tone sampled at 22.05 kHz (64 zero crossings of Hann-windowed sinc: error well below 1e-3 of the amplitude)."""
    sr0, sr1, f0 = 24000, 22050, 1234.5
    n = np.arange(24000)
    x = np.sin(2 * math.pi * f0 * n / sr0)[None]
    y = LO.resample(x, sr0, sr1)
    assert y.shape == (1, math.ceil(147 * 24000 / 160))
    m = np.arange(y.shape[1])
    want = np.sin(2 * math.pi * f0 * m / sr1)
    assert np.abs(y[0, 200:-200] - want[200:-200]).max() < 1e-3
    k, width, orig, new = LO.sinc_kernel(sr0, sr1)
    assert (orig, new, width, k.shape) == (160, 147, 74, (147, 308))
    assert abs(k.sum(axis=1) - 1.0).max() < 2e-2 * 160 / 147 + 0.1  # each phase is a unit-gain low-pass up to the rolloff scale


def test_logmel_against_torch_stft_and_filterbank_properties():
    torch.manual_seed(0)
    wav = 0.1 * torch.randn(2, 1, 24000, dtype=torch.float64)
    got = LO.wave_to_logmel(wav.numpy(), 22050, 22050, 256)  # no resample: isolates the mel / dB / framing part
    assert got.shape == (2, 96, 256)
    spec = torch.stft(wav[:, 0], 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024, periodic=True, dtype=torch.float64),
                      center=True, pad_mode="reflect", return_complex=True)
    fb = torch.from_numpy(LO.mel_filterbank(513, 20.0, 11025.0, 96, 22050))
    mel = torch.matmul((spec.abs() ** 2).transpose(1, 2), fb).transpose(1, 2).clamp_min(1e-8)
    db = 10 * torch.log10(mel.clamp_min(1e-10))
    db = torch.maximum(db, db.max() - 80.0)
    t = db.shape[-1]
    assert t == 94
    want = torch.cat([db, torch.full((2, 96, 256 - t), -80.0, dtype=torch.float64)], dim=2).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    # filterbank: triangles over consecutive HTK-mel points, non-negative, peak <= 1, every interior bin covered
    f = fb.numpy()
    assert f.min() >= 0.0 and f.max() <= 1.0 + 1e-12 and (f.sum(axis=1)[2:-1] > 0).all()
    centres = f.argmax(axis=0)
    assert (np.diff(centres) >= 0).all() and centres[0] >= 1
    # top_db: the floor follows the loudest bin of the WHOLE batch tensor
    loud = wav.clone(); loud[1] *= 1e5  # +100 dB: item 0 falls entirely under the shared floor
    g2 = LO.wave_to_logmel(loud.numpy(), 22050, 22050, 256)
    assert g2[:, :, :t].min() >= g2.max() - 80.0 - 1e-9 and g2[0, :, :t].min() == pytest.approx(g2.max() - 80.0, abs=1e-9)


def test_adapter_gradient_matches_oracle_finite_differences_through_the_top_db_floor():
    """gan_amd.lab1_adapter's autograd gradient against central differences of the float64 oracle, on a batch where one item lies
    entirely under the shared top_db floor: torchaudio's amplitude_to_DB keeps the graph through `amax() - top_db`, so the
    clamped bins' upstream gradient flows to the loudest bin of the batch (a detached floor - the round-2 code - drops it and
    fails this check; the forward values are the same either way)."""
    from gan_amd.lab1_adapter import wave_to_logmel_torch
    torch.manual_seed(3)
    wav = 0.05 * torch.randn(2, 1, 6000, dtype=torch.float64)
    wav[0] *= 1e-5                                   # -100 dB: item 0 sits on the floor everywhere
    w = torch.randn(2, 96, 64, dtype=torch.float64)  # upstream gradient
    v = torch.randn_like(wav)
    v[0] *= 1e-5
    x = wav.float().requires_grad_(True)
    out = wave_to_logmel_torch(x, 24000, 22050, 64)  # (the torch-op form of the adapter: the dB / top_db tail is shared with the HIP path)
    assert out.shape == (2, 96, 64)
    np.testing.assert_allclose(out.detach().numpy(), LO.wave_to_logmel(wav.numpy(), 24000, 22050, 64), atol=2e-3)
    on_floor = out.detach()[0] <= out.detach().max() - 80.0 + 1e-4
    assert int(on_floor.sum()) > 0.25 * on_floor.numel()  # (the frames past the signal's end are the -80 pad, not the floor)
    g, = torch.autograd.grad((out.double() * w).sum(), x)
    got = float((g.double() * v).sum())
    eps = 1e-6
    f = lambda t: float((torch.from_numpy(LO.wave_to_logmel(t.numpy(), 24000, 22050, 64)) * w).sum())
    want = (f(wav + eps * v) - f(wav - eps * v)) / (2 * eps)
    # the floor path alone: what a detached floor would lose
    floor_part = float(w[out.detach().double() <= float(out.detach().max()) - 80.0 + 1e-6].sum())
    assert abs(floor_part) > 1.0
    assert got == pytest.approx(want, rel=2e-3, abs=1e-3 * abs(want) + 1e-6), (got, want)
