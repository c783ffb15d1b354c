"""GPU: the mel-track product modules (convolutions on the HIP gathered-GEMM kernels) against the fixtures generated from the
reference's own classes (tests/golden/melgan_golden.npz) in fp32 mode, rtol 1e-4 on forwards and losses, fingerprint bounds on
gradients; and bf16 mode against the fp32 CPU oracle within the stated bound (outputs 3e-2 absolute on tanh / logit scale, parameter
gradient vectors cosine >= 0.99)."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_amd import lab3_models as PM
from gan_amd import lab3_train as PT
from gan_amd.optim import FusedAdamW
from oracle import mel_oracle as M
from oracle.procfill import fill_state_dict, proc_input

DEV = torch.device("cuda")
GOLD = np.load(Path(__file__).parent / "golden" / "melgan_golden.npz")


def fp(t):
    d = t.detach().double().cpu().reshape(-1)
    step = max(1, d.numel() // 8)
    return np.concatenate([[d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], d[::step][:8].numpy()])


def fp_close(got_t, want_fp, what, rtol=2e-4):
    got = fp(got_t)
    scale = max(abs(want_fp[1]), 1e-12)
    n = max(1, got_t.numel())
    if want_fp[1] / n < 1e-7:  # a gradient that is ZERO in exact arithmetic (a conv bias in front of a norm layer): rounding noise on both sides
        assert got[1] / n < 1e-6, (what, "noise-sized in the reference, not here", got[1] / n)
        return
    assert abs(got[0] - want_fp[0]) <= rtol * scale + 1e-7, (what, "sum", got[0], want_fp[0])
    assert abs(got[1] - want_fp[1]) <= rtol * scale + 1e-7, (what, "abs-sum", got[1], want_fp[1])
    assert abs(got[2] - want_fp[2]) <= 2 * rtol * max(want_fp[2], 1e-12) + 1e-9, (what, "sq-sum", got[2], want_fp[2])
    assert np.all(np.abs(got[3:] - want_fp[3:]) <= rtol * np.abs(want_fp[3:]) + 50 * rtol * scale / n + 1e-7), (what, "samples", got[3:], want_fp[3:])


def close(got, want, rtol=1e-4, atol=1e-6, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    assert got.shape == want.shape and np.all(err <= atol + rtol * np.abs(want)), (what, float(err.max()), float(np.abs(want).max()))


def load_filled(mod, seed):
    mod.load_state_dict(fill_state_dict(mod.state_dict(), seed))
    return mod.to(DEV).train()


DEC_CASES = {"default": {}, "ps_batch": dict(upsample="pixelshuffle", norm="batch"), "nearest": dict(upsample="nearest"),
             "mrf": dict(mrf=True, base_channels=64), "small": dict(zc_dim=16, cond_dim=8, n_mels=16, n_frames=32, base_channels=32)}


@pytest.mark.parametrize("tag", list(DEC_CASES))
def test_decoder_matches_the_reference(tag):
    kw = DEC_CASES[tag]
    G = load_filled(PM.ReconstructionDecoder(**kw), 11)
    zc = proc_input((2, kw.get("zc_dim", 128)), f"melgan.zc.{tag}", 0).to(DEV)
    v = proc_input((2, kw.get("cond_dim", 160)), f"melgan.v.{tag}", 0).to(DEV)
    mel = G(zc, v)
    assert tuple(mel.shape) == (2, G.n_mels, G.n_frames) and mel.dtype == torch.float32
    fp_close(mel, GOLD[f"dec/{tag}/out_fp"], f"{tag} output", rtol=1e-4)
    close(mel[:, ::5, ::17].detach().cpu().numpy(), GOLD[f"dec/{tag}/out_slice"], atol=2e-6, what=f"{tag} slice")
    if f"dec/{tag}/loss" in GOLD:
        tgt = proc_input(tuple(mel.shape), f"melgan.tgt.{tag}", 0).to(DEV)
        loss = ((mel - tgt) ** 2).mean() + 0.1 * mel.abs().mean()
        close(float(loss.detach()), GOLD[f"dec/{tag}/loss"], what="loss")
        names = [n for n, _ in G.named_parameters()]
        for n, g in zip(names, torch.autograd.grad(loss, [p for _, p in G.named_parameters()])):
            fp_close(g, GOLD[f"dec/{tag}/grad/{n}"], f"{tag} grad {n}", rtol=3e-4)
    if tag == "ps_batch":
        for n, b in G.named_buffers():
            if b.is_floating_point():
                close(b.cpu().numpy(), GOLD[f"dec/{tag}/buf/{n}"], what=n)


DISC_CASES = {"plain": (lambda: PM.MelDiscriminator(cond_dim=0), False), "proj": (lambda: PM.MelDiscriminator(cond_dim=160), True),
              "proj_small": (lambda: PM.MelDiscriminator(base_channels=16, cond_dim=160), True),
              "multiscale": (lambda: PM.MultiScaleMelDiscriminator(cond_dim=160, base_channels=16), True),
              "subband": (lambda: PM.SubBandMelDiscriminator(cond_dim=160, base_channels=16), True),
              "multiperiod": (lambda: PM.MultiPeriodMelDiscriminator(cond_dim=160, base_channels=16), True),
              "hybrid": (lambda: PM.HybridMelDiscriminator(cond_dim=160), True)}


@pytest.mark.parametrize("tag", list(DISC_CASES))
def test_discriminators_match_the_reference(tag):
    mk, use_cond = DISC_CASES[tag]
    D = load_filled(mk(), 12)
    mel = proc_input((3, 96, 256), "melgan.d.mel", 0, 0.8).to(DEV).requires_grad_(True)
    cond = proc_input((3, 160), "melgan.d.cond", 0).to(DEV)
    logit, feat = D(mel, cond if use_cond else None, return_features=True)
    close(logit.detach().cpu().numpy(), GOLD[f"disc/{tag}/logit"], atol=2e-6, what="logit")
    fp_close(feat, GOLD[f"disc/{tag}/feat_fp"], "features", rtol=1e-4)
    loss = F.softplus(-logit).mean() + 0.01 * feat.pow(2).mean()
    close(float(loss.detach()), GOLD[f"disc/{tag}/loss"], what="loss")
    names = [n for n, _ in D.named_parameters()]
    grads = torch.autograd.grad(loss, [mel] + [p for _, p in D.named_parameters()])
    fp_close(grads[0], GOLD[f"disc/{tag}/grad_x_fp"], "grad x", rtol=3e-4)
    # kink-tolerant: a LeakyReLU whose pre-activation is ~0 may take the other slope here, which changes the input gradient inside
    # that unit's receptive field by a finite amount (not an error of either side): at most 5 % of the sampled entries may differ
    # by more than 3e-4, and the slice agrees to 2 % in relative L2
    want_slice = GOLD[f"disc/{tag}/grad_x_slice"]
    got_slice = grads[0][:, ::9, ::31].cpu().numpy().astype(np.float64)
    bad = np.abs(got_slice - want_slice) > 3e-4 * np.abs(want_slice) + 1e-4 * np.abs(want_slice).max()
    assert bad.mean() <= 0.05, ("grad x slice", float(bad.mean()))
    assert np.linalg.norm(got_slice - want_slice) <= 2e-2 * np.linalg.norm(want_slice), "grad x slice (L2)"
    for n, g in zip(names, grads[1:]):
        fp_close(g, GOLD[f"disc/{tag}/grad/{n}"], f"grad {n}", rtol=2e-3)  # (2e-3: the same kinks; measured 8e-4 on the 16-channel net's BatchNorm gains, <= 4e-4 elsewhere)
    for n, b in D.named_buffers():
        if b.is_floating_point():
            fp_close(b, GOLD[f"disc/{tag}/buf/{n}"], f"buffer {n}", rtol=1e-4)
    if tag == "proj":
        D.eval()
        close(D(mel.detach(), cond).detach().cpu().numpy(), GOLD["disc/proj/eval_logit"], atol=2e-6, what="eval")


def _noise_bias(name):
    return name in ("up1.0.bias", "up2.0.bias", "up3.0.bias", "D.net.2.bias", "D.net.5.bias", "D.net.8.bias")


@pytest.mark.parametrize("tag,gan_loss,clip", [("bce", "bce", 0.0), ("hinge_clip", "hinge", 1.5)])
def test_stage1_iterations_match_the_reference(tag, gan_loss, clip):
    """Three passes of train_stage1's loop body through mel_stage1_iteration (fp32 mode, FusedAdamW as Adam) against the sequence the
    reference's pieces produced: losses to 1e-4 / 3e-4 / 3e-3 by iteration (Adam's first steps are +-lr whatever the gradient's
    size), parameters by fingerprint, BatchNorm buffers; biases in front of a norm layer (true gradient zero: pure rounding noise
    through Adam) within 2 lr per step."""
    G = load_filled(PM.ReconstructionDecoder(zc_dim=32, cond_dim=24, n_mels=32, n_frames=64, base_channels=64), 21)
    D = load_filled(PM.MelDiscriminator(base_channels=16, cond_dim=24), 22)
    cfg = PT.StageTrainConfig(stage_name="stage1", epochs=1, gan_loss=gan_loss, g_grad_clip_norm=clip, d_grad_clip_norm=clip,
                              mrstft_resolutions=((16, 4, 16), (32, 8, 32), (64, 16, 64)), weights=PT.TrainWeights(content=0.0, style=0.0))
    og = FusedAdamW(G.parameters(), lr=cfg.lr_g, betas=cfg.betas, weight_decay=0.0)
    od = FusedAdamW(D.parameters(), lr=cfg.lr_d, betas=cfg.betas, weight_decay=0.0)
    for it in range(3):
        real = proc_input((4, 32, 64), f"melgan.it{it}.real", 0, 0.8).to(DEV)
        zc = proc_input((4, 32), f"melgan.it{it}.zc", 0).to(DEV)
        cond = proc_input((4, 24), f"melgan.it{it}.cond", 0).to(DEV)
        out = PT.mel_stage1_iteration(cfg, G, D, og, od, real, zc, cond, it)
        rt = (1e-4, 3e-4, 3e-3)[it]
        for k in ("loss_d", "loss_g", "loss_adv", "loss_l1", "loss_continuity"):
            close(float(out[k]), GOLD[f"iter/{tag}/{it}/{k}"], rtol=rt, what=f"it{it} {k}")
        for n, p in list(G.named_parameters()) + [("D." + n, p) for n, p in D.named_parameters()]:
            want = GOLD[f"iter/{tag}/{it}/param/{n}"]
            if _noise_bias(n):
                got = fp(p)
                tol = p.numel() * 2.0 * cfg.lr_g * (it + 1) + 1e-6
                assert abs(got[0] - want[0]) <= tol and abs(got[1] - want[1]) <= tol, (n, got[:2], want[:2])
            elif it == 0:
                fp_close(p, want, f"it{it} {n}", rtol=1e-3)
            else:
                # from the second step on the parameters inherit the sign noise above through the gradients (every early Adam step is
                # +-lr): up to half of a tensor's entries 2 lr apart per step taken, on top of the relative bound
                got = fp(p)
                tol = 1e-3 * max(abs(want[1]), 1e-12) + max(0.5 * p.numel(), 1.0) * 2.0 * cfg.lr_g * it + 1e-7
                assert abs(got[0] - want[0]) <= tol and abs(got[1] - want[1]) <= tol, (f"it{it} {n}", got[:2], want[:2], tol)
        for n, b in D.named_buffers():
            if b.is_floating_point():
                if it == 0:
                    # (a running mean contains its conv's bias -- one of the noise-stepped parameters -- as of the D step just taken)
                    atol = 2.0 * cfg.lr_d + 2e-6 if n.endswith("running_mean") else 1e-6
                    close(b.cpu().numpy(), GOLD[f"iter/{tag}/{it}/dbuf/{n}"], rtol=1e-3, atol=atol, what=f"it{it} buffer {n}")
                else:  # later: every weight of the conv in front has taken +-lr steps with noise-decided signs (measured: 4e-3 on 0.2)
                    close(b.cpu().numpy(), GOLD[f"iter/{tag}/{it}/dbuf/{n}"], rtol=2e-2, atol=2e-2, what=f"it{it} buffer {n}")


def test_bf16_mode_stays_within_the_stated_bound():
    """bf16 MFMA operands, fp32 accumulation and parameters: decoder output within 3e-2 (tanh scale), discriminator logits within
    5e-2 absolute, parameter-gradient vectors with cosine >= 0.99 against the fp32 CPU oracle on the same procedural state."""
    dcfg = M.DecoderCfg()
    G = load_filled(PM.ReconstructionDecoder(precision="bf16"), 11)
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    zc, v = proc_input((2, 128), "melgan.zc.default", 0), proc_input((2, 160), "melgan.v.default", 0)
    leaf = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
    want = M.decoder_forward(leaf, zc, v, dcfg)
    tgt = proc_input(tuple(want.shape), "melgan.tgt.default", 0)
    wl = ((want - tgt) ** 2).mean() + 0.1 * want.abs().mean()
    wg = torch.cat([g.reshape(-1) for g in torch.autograd.grad(wl, list(leaf.values()))]).double()
    got = G(zc.to(DEV), v.to(DEV))
    assert float((got.cpu() - want.detach()).abs().max()) <= 3e-2
    gl = ((got - tgt.to(DEV)) ** 2).mean() + 0.1 * got.abs().mean()
    gg = torch.cat([g.reshape(-1) for g in torch.autograd.grad(gl, [p for _, p in G.named_parameters()])]).double().cpu()
    cos = float(torch.dot(gg, wg) / (gg.norm() * wg.norm()))
    print(f"decoder bf16 gradient cosine {cos:.5f}, relative L2 error {float((gg - wg).norm() / wg.norm()):.4f}")
    assert cos >= 0.99
    D = load_filled(PM.MelDiscriminator(cond_dim=160, precision="bf16"), 12)
    dsd = {k: t.detach().cpu() for k, t in D.state_dict().items()}
    mel, cond = proc_input((3, 96, 256), "melgan.d.mel", 0, 0.8), proc_input((3, 160), "melgan.d.cond", 0)
    dleaf = {k: (t.clone().requires_grad_(True) if t.is_floating_point() and "running_" not in k else t) for k, t in dsd.items()}
    wlog, wfeat = M.mel_disc_forward(dleaf, "", mel, cond, {})
    wloss = F.softplus(-wlog).mean() + 0.01 * wfeat.pow(2).mean()
    names = [n for n, _ in D.named_parameters()]
    wgd = torch.cat([g.reshape(-1) for g in torch.autograd.grad(wloss, [dleaf[n] for n in names])]).double()
    glog, gfeat = D(mel.to(DEV), cond.to(DEV), return_features=True)
    assert float((glog.cpu() - wlog.detach()).abs().max()) <= 5e-2
    gloss = F.softplus(-glog).mean() + 0.01 * gfeat.pow(2).mean()
    ggd = torch.cat([g.reshape(-1) for g in torch.autograd.grad(gloss, [p for _, p in D.named_parameters()])]).double().cpu()
    cosd = float(torch.dot(ggd, wgd) / (ggd.norm() * wgd.norm()))
    print(f"discriminator bf16 gradient cosine {cosd:.5f}, relative L2 error {float((ggd - wgd).norm() / wgd.norm()):.4f}")
    assert cosd >= 0.99


def test_train_stage1_runs_checkpoints_and_resumes(tmp_path):
    """train_stage1 with the reference's arguments on a toy loader: history rows with the reference's keys, a checkpoint with its
    keys that a second call resumes from (no further epochs to run), the content / style terms through a caller-supplied encoder."""
    torch.manual_seed(5)
    G = PM.ReconstructionDecoder(zc_dim=16, cond_dim=8, n_mels=32, n_frames=64, base_channels=32).to(DEV)
    D = PM.MelDiscriminator(base_channels=16, cond_dim=8).to(DEV)

    class ToyEncoder(torch.nn.Module):  # stands in for the caller's FrozenLab1Encoder (lab3_bridge.py): a differentiable map of the dB mel
        def __init__(self):
            super().__init__()
            self.c, self.s = torch.nn.Linear(32, 16), torch.nn.Linear(32, 16)

        def forward_log_mel_tensor(self, mel_db):
            m = mel_db.mean(dim=-1) / 80.0
            return {"z_content": self.c(m), "z_style": self.s(m)}

    enc = ToyEncoder().to(DEV).requires_grad_(False)
    batches = [{"mel_norm": torch.rand(4, 32, 64) * 2 - 1, "z_content": torch.randn(4, 16), "z_style": torch.randn(4, 16),
                "genre_idx": torch.randint(0, 3, (4,))} for _ in range(3)]
    cond_bank = torch.randn(3, 8)
    cfg = PT.StageTrainConfig(stage_name="stage1", epochs=2, gan_loss="hinge", d_step_period=2, g_grad_clip_norm=5.0,
                              mrstft_resolutions=((32, 8, 32), (64, 16, 64)))
    ckpt = tmp_path / "stage1.pt"
    hist = PT.train_stage1(G, D, enc, batches, cond_bank, DEV, cfg, ckpt)
    assert len(hist) == 2 and hist[-1]["epoch"] == 2 and hist[0]["stage"] == "stage1"
    assert set(hist[0]) == {"loss_g", "loss_d", "loss_r1", "loss_adv", "loss_l1", "loss_content", "loss_style", "loss_continuity", "loss_mrstft",
                            "stage", "epoch", "d_steps_taken", "d_steps_skipped", "d_step_rate"}
    assert hist[0]["d_steps_taken"] == 2.0 and hist[0]["d_steps_skipped"] == 1.0 and hist[1]["d_steps_taken"] == 1.0  # steps 0..5, period 2
    assert all(np.isfinite(v) for row in hist for v in row.values() if isinstance(v, float))
    assert hist[0]["loss_content"] > 0 and hist[0]["loss_style"] > 0
    saved = torch.load(ckpt, map_location="cpu", weights_only=True)
    assert set(saved) == {"epoch", "generator", "discriminator", "opt_g", "opt_d", "meta"} and saved["epoch"] == 2
    assert set(saved["generator"]) == set(G.state_dict())
    assert PT.train_stage1(G, D, enc, batches, cond_bank, DEV, cfg, ckpt, resume=True) == []
    with pytest.raises(ValueError):  # content weight without an encoder is an error, not a silent drop
        PT.train_stage1(G, D, None, batches, cond_bank, DEV, cfg, tmp_path / "x.pt")
