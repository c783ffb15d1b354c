"""CPU: the mel-track oracle (oracle/mel_oracle.py) against tests/golden/melgan_golden.npz, which make_golden_melgan.py generated
from the reference's own ReconstructionDecoder / MelDiscriminator classes and GAN loss helpers (lab 3/src/lab3_models.py,
lab3_train.py).  rtol 1e-4 unless a line says otherwise."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mel_oracle as M
from oracle.procfill import fill_state_dict, proc_input
from tests.mel_shapes import decoder_state_shapes, disc_state_shapes, wrapper_state_shapes

GOLD = np.load(Path(__file__).parent / "golden" / "melgan_golden.npz")


def fp(t):
    d = t.detach().double().reshape(-1)
    step = max(1, d.numel() // 8)
    return np.concatenate([[d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], d[::step][:8].numpy()])


def close(got, want, rtol=1e-4, atol=1e-6, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    assert np.all(err <= atol + rtol * np.abs(want)), (what, float(err.max()), float(np.abs(want).max()))


def fp_close(got_t, want_fp, what, rtol=1e-4):
    got = fp(got_t)
    scale = max(abs(want_fp[1]), 1e-12)  # the abs-sum sets the scale of the signed sum and of the samples
    n = max(1, got_t.numel())
    assert abs(got[0] - want_fp[0]) <= rtol * scale + 1e-7, (what, "sum", got[0], want_fp[0])
    assert abs(got[1] - want_fp[1]) <= rtol * scale + 1e-7, (what, "abs-sum", got[1], want_fp[1])
    assert abs(got[2] - want_fp[2]) <= 2 * rtol * max(want_fp[2], 1e-12) + 1e-9, (what, "sq-sum")
    assert np.all(np.abs(got[3:] - want_fp[3:]) <= rtol * np.abs(want_fp[3:]) + 50 * rtol * scale / n + 1e-7), (what, "samples")


DEC_CASES = {"default": {}, "ps_batch": dict(upsample="pixelshuffle", norm="batch"), "nearest": dict(upsample="nearest"),
             "mrf": dict(mrf=True, base_channels=64), "small": dict(zc_dim=16, cond_dim=8, n_mels=16, n_frames=32, base_channels=32)}


@pytest.mark.parametrize("tag", list(DEC_CASES))
def test_decoder_forward_and_gradients(tag):
    cfg = M.DecoderCfg(**DEC_CASES[tag])
    sd = fill_state_dict({k: torch.zeros(s, dtype=(torch.long if "num_batches" in k else torch.float32)) for k, s in decoder_state_shapes(cfg).items()}, 11)
    leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    zc = proc_input((2, cfg.zc_dim), f"melgan.zc.{tag}", 0)
    v = proc_input((2, cfg.cond_dim), f"melgan.v.{tag}", 0)
    bufs = {}
    mel = M.decoder_forward(leaf, zc, v, cfg, buffers=bufs)
    assert tuple(mel.shape) == (2, cfg.n_mels, cfg.n_frames)
    fp_close(mel, GOLD[f"dec/{tag}/out_fp"], f"{tag} output")
    close(mel[:, ::5, ::17].detach().numpy(), GOLD[f"dec/{tag}/out_slice"], what=f"{tag} slice")
    if f"dec/{tag}/loss" in GOLD:
        tgt = proc_input(tuple(mel.shape), f"melgan.tgt.{tag}", 0)
        loss = ((mel - tgt) ** 2).mean() + 0.1 * mel.abs().mean()
        close(float(loss), GOLD[f"dec/{tag}/loss"], what="loss")
        names = [k for k, t in leaf.items() if t.is_floating_point() and t.requires_grad]
        for n, g in zip(names, torch.autograd.grad(loss, [leaf[k] for k in names])):
            fp_close(g, GOLD[f"dec/{tag}/grad/{n}"], f"{tag} grad {n}", rtol=2e-4)
    if tag == "ps_batch":
        for k, b in bufs.items():
            close(b.numpy(), GOLD[f"dec/{tag}/buf/{k}"], what=k)


DISC_FWD = {
    "plain": (lambda sd, x, c, b: M.mel_disc_forward(sd, "", x, None, b), dict(kind="single", base=64, cond_dim=0)),
    "proj": (lambda sd, x, c, b: M.mel_disc_forward(sd, "", x, c, b), dict(kind="single", base=64, cond_dim=160)),
    "proj_small": (lambda sd, x, c, b: M.mel_disc_forward(sd, "", x, c, b), dict(kind="single", base=16, cond_dim=160)),
    "multiscale": (lambda sd, x, c, b: M.multiscale_forward(sd, "", x, c, 3, b), dict(kind="multiscale", base=16, cond_dim=160)),
    "subband": (lambda sd, x, c, b: M.subband_forward(sd, "", x, c, 32, 32, b), dict(kind="subband", base=16, cond_dim=160)),
    "multiperiod": (lambda sd, x, c, b: M.multiperiod_forward(sd, "", x, c, (1, 2, 3, 5), b), dict(kind="multiperiod", base=16, cond_dim=160)),
    "hybrid": (lambda sd, x, c, b: M.hybrid_forward(sd, x, c, 3, (1, 2, 3, 5), b), dict(kind="hybrid", base=64, cond_dim=160)),
}


@pytest.mark.parametrize("tag", list(DISC_FWD))
def test_discriminators_forward_gradients_and_buffers(tag):
    fwd, spec = DISC_FWD[tag]
    shapes = wrapper_state_shapes(**spec)
    sd = fill_state_dict({k: torch.zeros(s, dtype=(torch.long if "num_batches" in k else torch.float32)) for k, s in shapes.items()}, 12)
    leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    mel = proc_input((3, 96, 256), "melgan.d.mel", 0, 0.8).requires_grad_(True)
    cond = proc_input((3, 160), "melgan.d.cond", 0)
    bufs = {}
    logit, feat = fwd(leaf, mel, cond, bufs)
    close(logit.detach().numpy(), GOLD[f"disc/{tag}/logit"], what="logit")
    fp_close(feat, GOLD[f"disc/{tag}/feat_fp"], "features")
    loss = F.softplus(-logit).mean() + 0.01 * feat.pow(2).mean()
    close(float(loss), GOLD[f"disc/{tag}/loss"], what="loss")
    names = [k for k, t in leaf.items() if t.is_floating_point() and t.requires_grad]
    grads = torch.autograd.grad(loss, [mel] + [leaf[k] for k in names])
    fp_close(grads[0], GOLD[f"disc/{tag}/grad_x_fp"], "grad x", rtol=2e-4)
    close(grads[0][:, ::9, ::31].numpy(), GOLD[f"disc/{tag}/grad_x_slice"], rtol=2e-4, atol=1e-9, what="grad x slice")
    for n, g in zip(names, grads[1:]):
        fp_close(g, GOLD[f"disc/{tag}/grad/{n}"], f"grad {n}", rtol=3e-4)
    for k, b in bufs.items():
        if b.is_floating_point():
            fp_close(b, GOLD[f"disc/{tag}/buf/{k}"], f"buffer {k}")
    if tag == "proj":
        sd2 = dict(sd)
        sd2.update(bufs)
        close(M.mel_disc_forward(sd2, "", mel.detach(), cond, None, training=False)[0].numpy(), GOLD["disc/proj/eval_logit"], what="eval")


def test_gan_losses():
    a = proc_input((7,), "melgan.loss.a", 0, 2.0)
    b = proc_input((7,), "melgan.loss.b", 0, 2.0)
    close(float(M.bce_logits(a, True)), GOLD["loss/bce_real"])
    close(float(M.bce_logits(a, False, 0.9, 0.1)), GOLD["loss/bce_fake_smooth"])
    for key in ("bce", "hinge"):
        close(float(M.gan_d_loss(key, a, b, 0.9, 0.05)), GOLD[f"loss/d_{key}"])
        close(float(M.gan_g_loss(key, b, 0.95, 0.05)), GOLD[f"loss/g_{key}"])
    with pytest.raises(ValueError):
        M.gan_d_loss("wgan", a, b)


def _bias_in_front_of_a_norm(name: str) -> bool:
    """Conv biases that an InstanceNorm / BatchNorm removes again (decoder up-blocks, D's second to fourth conv): their true gradient
    is ZERO, what backward returns is rounding noise (~1e-9), and Adam turns noise into +-lr steps whose signs no two
    implementations share."""
    return name in ("up1.0.bias", "up2.0.bias", "up3.0.bias", "D.net.2.bias", "D.net.5.bias", "D.net.8.bias")


def _noise_step_close(p, want_fp, it, lr, name):
    got = fp(p)
    tol = p.numel() * 2.0 * lr * (it + 1) + 1e-6  # every entry at most 2 lr apart per Adam step taken
    assert abs(got[0] - want_fp[0]) <= tol and abs(got[1] - want_fp[1]) <= tol, (name, got[:2], want_fp[:2], tol)


@pytest.mark.parametrize("tag,gan_loss,clip", [("bce", "bce", 0.0), ("hinge_clip", "hinge", 1.5)])
def test_stage1_iteration_sequence(tag, gan_loss, clip):
    """Three iterations of train_stage1's loop body: losses to 1e-4 / 2e-4 / 5e-4 (iteration 1 / 2 / 3: Adam's first steps are
    +-lr whatever the gradient's size, so 1e-7 differences grow), first-iteration gradients, parameters and BatchNorm buffers."""
    dcfg = M.DecoderCfg(zc_dim=32, cond_dim=24, n_mels=32, n_frames=64, base_channels=64)
    g_sd = fill_state_dict({k: torch.zeros(s) for k, s in decoder_state_shapes(dcfg).items()}, 21)
    d_sd = fill_state_dict({k: torch.zeros(s, dtype=(torch.long if "num_batches" in k else torch.float32))
                            for k, s in disc_state_shapes(16, 24).items()}, 22)
    cfg = M.MelStepCfg(gan_loss=gan_loss, g_grad_clip_norm=clip, d_grad_clip_norm=clip, decoder=dcfg,
                       mrstft_resolutions=((16, 4, 16), (32, 8, 32), (64, 16, 64)))
    og, od = M.adam_state(cfg.lr_g, cfg.betas), M.adam_state(cfg.lr_d, cfg.betas)
    for it in range(3):
        real = proc_input((4, 32, 64), f"melgan.it{it}.real", 0, 0.8)
        zc = proc_input((4, 32), f"melgan.it{it}.zc", 0)
        cond = proc_input((4, 24), f"melgan.it{it}.cond", 0)
        out = M.stage1_iteration(g_sd, d_sd, og, od, real, zc, cond, it, cfg, record=(it == 0))
        rt = (1e-4, 2e-4, 5e-4)[it]
        for k in ("loss_d", "loss_g", "loss_adv", "loss_l1", "loss_continuity"):
            close(out[k], GOLD[f"iter/{tag}/{it}/{k}"], rtol=rt, what=f"it{it} {k}")
        if it == 0:
            for n, g in out["d_grads"].items():
                fp_close(g, GOLD[f"iter/{tag}/0/d_grad/{n}"], f"d grad {n}", rtol=3e-4)
            for n, g in out["g_grads"].items():
                fp_close(g, GOLD[f"iter/{tag}/0/g_grad/{n}"], f"g grad {n}", rtol=3e-4)
        for n, p in g_sd.items():
            if _bias_in_front_of_a_norm(n):
                _noise_step_close(p, GOLD[f"iter/{tag}/{it}/param/{n}"], it, cfg.lr_g, n)
            else:
                fp_close(p, GOLD[f"iter/{tag}/{it}/param/{n}"], f"it{it} G {n}", rtol=5e-4)
        for n, p in d_sd.items():
            if p.is_floating_point() and "running_" not in n:
                if _bias_in_front_of_a_norm("D." + n):
                    _noise_step_close(p, GOLD[f"iter/{tag}/{it}/param/D.{n}"], it, cfg.lr_d, n)
                else:
                    fp_close(p, GOLD[f"iter/{tag}/{it}/param/D.{n}"], f"it{it} D {n}", rtol=5e-4)
            elif p.is_floating_point():
                # (a running mean contains its conv's bias, one of the noise-stepped parameters above: up to 2 lr per step taken)
                atol = 2.0 * cfg.lr_d * it + 2e-6 if n.endswith("running_mean") else 1e-6
                close(p.numpy(), GOLD[f"iter/{tag}/{it}/dbuf/{n}"], rtol=5e-4, atol=atol, what=f"it{it} buffer {n}")
