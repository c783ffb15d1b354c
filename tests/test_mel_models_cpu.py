"""CPU: the mel-track product modules (gan_amd/lab3_models.py, lab3_train.py) expose the reference's state-dict keys, shapes and
constructor errors, and refuse to run without a GPU (no CPU fallback)."""
import pytest
import torch

from gan_amd import lab3_models as PM
from gan_amd import lab3_train as PT
from oracle import mel_oracle as M
from tests.mel_shapes import decoder_state_shapes, wrapper_state_shapes


@pytest.mark.parametrize("kw", [{}, dict(upsample="pixelshuffle", norm="batch"), dict(upsample="nearest"), dict(mrf=True, base_channels=64),
                                dict(zc_dim=16, cond_dim=8, n_mels=16, n_frames=32, base_channels=32)])
def test_decoder_state_dict_matches_the_reference_layout(kw):
    G = PM.ReconstructionDecoder(**kw)
    want = decoder_state_shapes(M.DecoderCfg(**kw))
    got = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    assert list(got) == list(want), (list(got), list(want))  # same keys in the same (registration) order
    assert got == want


@pytest.mark.parametrize("mk,spec", [
    (lambda: PM.MelDiscriminator(cond_dim=0), dict(kind="single", base=64, cond_dim=0)),
    (lambda: PM.MelDiscriminator(base_channels=16, cond_dim=160), dict(kind="single", base=16, cond_dim=160)),
    (lambda: PM.MultiScaleMelDiscriminator(cond_dim=160, base_channels=16), dict(kind="multiscale", base=16, cond_dim=160)),
    (lambda: PM.SubBandMelDiscriminator(cond_dim=160, base_channels=16), dict(kind="subband", base=16, cond_dim=160)),
    (lambda: PM.MultiPeriodMelDiscriminator(cond_dim=160, base_channels=16), dict(kind="multiperiod", base=16, cond_dim=160)),
    (lambda: PM.HybridMelDiscriminator(cond_dim=160), dict(kind="hybrid", base=64, cond_dim=160))])
def test_discriminator_state_dicts_match_the_reference_layout(mk, spec):
    got = {k: tuple(v.shape) for k, v in mk().state_dict().items()}
    want = wrapper_state_shapes(**spec)
    assert set(got) == set(want) and got == {k: want[k] for k in got}


def test_constructor_errors_and_unsupported_options():
    with pytest.raises(ValueError):
        PM.ReconstructionDecoder(n_mels=90)
    with pytest.raises(ValueError):
        PM.ReconstructionDecoder(upsample="bilinear")
    with pytest.raises(ValueError):
        PM.ReconstructionDecoder(norm="layer")
    with pytest.raises(ValueError):
        PM.MultiScaleMelDiscriminator(num_scales=0)
    sn_keys = set(PM.MelDiscriminator(base_channels=16, cond_dim=8, spectral_norm=True).state_dict())  # torch's spectral_norm layout
    assert {"net.0.weight_orig", "net.0.weight_u", "net.0.weight_v", "head.weight_orig", "cond_proj.weight_u"} <= sn_keys and "net.0.weight" not in sn_keys
    with pytest.raises(ValueError):
        PT._gan_d_loss("wgan", torch.zeros(2), torch.zeros(2), 1.0, 0.0)
    cfg = PT.StageTrainConfig(stage_name="stage1", epochs=1)
    assert cfg.gan_loss == "bce" and cfg.betas == (0.5, 0.999) and cfg.weights.recon_l1 == 8.0 and cfg.mrstft_resolutions[0] == (64, 16, 64)


def test_no_cpu_fallback():
    G = PM.ReconstructionDecoder(zc_dim=16, cond_dim=8, n_mels=16, n_frames=32, base_channels=32)
    with pytest.raises(Exception):
        G(torch.zeros(1, 16), torch.zeros(1, 8))
    D = PM.MelDiscriminator(base_channels=16)
    with pytest.raises(Exception):
        D(torch.zeros(1, 32, 64))


def test_chunk_encoder_layout_and_helpers():
    """gan_amd.lab3_bridge: the reference ChunkEncoder's parameter names (lab3_bridge.py:37-60), no CPU fallback, and the array helpers."""
    import numpy as np
    from gan_amd import lab3_bridge as PB
    enc = PB.ChunkEncoder(n_sources=4, z_dim=64)
    keys = list(enc.state_dict())
    assert keys[:2] == ["backbone.0.weight", "backbone.0.bias"] and "backbone.7.running_var" in keys and "content_style_adv.2.bias" in keys
    assert tuple(enc.state_dict()["backbone.6.weight"].shape) == (128, 64, 3, 3) and tuple(enc.state_dict()["style_cls.weight"].shape) == (4, 64)
    with pytest.raises(Exception):
        enc(torch.zeros(1, 96, 32))
    db = np.linspace(-100.0, 10.0, 12, dtype=np.float32).reshape(3, 4)
    n = PB.normalize_log_mel(db)
    assert n.min() == -1.0 and n.max() == 1.0 and n.dtype == np.float32
    np.testing.assert_allclose(PB.denormalize_log_mel(torch.from_numpy(n)).numpy(), np.clip(db, -80.0, 0.0), atol=1e-5)
    assert PB.fix_log_mel_frames(db, 2).tolist() == db[:, 1:3].tolist()
    padded = PB.fix_log_mel_frames(db, 6)
    assert padded.shape == (3, 6) and padded[0, 5] == -80.0 and padded[1, 3] == db[1, 3]
    with pytest.raises(ValueError):
        PB.fix_log_mel_frames(db[0], 4)
    d = PB.descriptor32_from_logmel(np.random.default_rng(0).normal(size=(96, 40)).astype(np.float32))
    assert d.shape == (32,) and abs(float(d.mean())) < 1e-6 and abs(float(d.std()) - 1.0) < 1e-3
    x = torch.ones(3, requires_grad=True)
    PB.grad_reverse(x, 0.5).sum().backward()
    assert x.grad.tolist() == [-0.5, -0.5, -0.5]
