"""CPU oracle for the lab3 codec-track adversarial training step.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file; nothing under gan_amd/ does, and the product path has no CPU fallback.

It is an independent restatement, in stock torch CPU fp32 ops, of the reference algorithm:

  * generator G ............ /root/reference/lab 3/src/lab3_codec_models.py:11-119
  * discriminator D ........ lab3_codec_models.py:122-168
  * hinge / FM losses ...... lab3_codec_models.py:171-215
  * MR-STFT loss ........... /root/reference/lab 3/src/lab3_codec_train.py:148-176
  * codec style judge ...... /root/reference/lab 3/src/lab3_codec_judge.py:83-145 and the style terms of the G step,
    lab3_codec_train.py:480-494 (pinned by tests/golden/judge_golden.npz, made by tests/golden/make_golden_judge.py)
  * training iteration ..... lab3_codec_train.py:404-591 (north-star loss subset), clip + AdamW :302-311,446-448,589-591
  * frozen SEANet decoder .. third-party `transformers` 5.15.0 (unpinned in lab 3/requirements.txt:6),
    models/encodec/modeling_encodec.py:76-176 (EncodecConv1d), :179-233 (EncodecConvTranspose1d),
    :236-249 (EncodecLSTM), :252-283 (EncodecResnetBlock), :316-347 (EncodecDecoder); called by the
    reference at lab3_codec_bridge.py:135-146.

Pinned against tests/golden/codec_golden.npz, which tests/golden/make_golden.py produced by running the
reference's own modules (and the HF decoder) in the build container with procedurally filled weights.
The reference has no tests or fixtures of its own for this path, so those vectors
are the only pin; the pretrained EnCodec weights were never available, so decoder parity is pinned on
the architecture with procedural weights.

Everything is functional over plain state dicts with the reference's key names, so the same dict loads
into the reference modules, into this oracle and into gan_amd's modules.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

DEFAULT_RESOLUTIONS = ((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048))


# ------------------------------------------------------------------------------------------------
# generator (lab3_codec_models.py:11-119)
def g_dilations(n_blocks: int) -> List[int]:
    d = [1, 2, 4, 8, 1, 2, 4, 8, 1, 2]
    if int(n_blocks) != len(d):
        d = [2 ** (i % 4) for i in range(int(n_blocks))]
    return d


def _film(sd: SD, pre: str, x: Tensor, cond: Tensor) -> Tensor:
    ab = F.linear(cond, sd[pre + "proj.0.weight"], sd[pre + "proj.0.bias"])
    ab = F.silu(ab)
    ab = F.linear(ab, sd[pre + "proj.2.weight"], sd[pre + "proj.2.bias"])
    c = ab.shape[1] // 2
    a, b = ab[:, :c], ab[:, c:]
    return x * (1.0 + a.unsqueeze(-1)) + b.unsqueeze(-1)


def generator_forward(sd: SD, q_src: Tensor, z_content: Tensor, z_style_tgt: Tensor,
                      noise: Optional[Tensor] = None, residual_scale: float = 0.5,
                      direct_output: bool = False, noise_dim: int = 32,
                      style_keep: Optional[Tensor] = None, style_jitter: Optional[Tensor] = None) -> Tensor:
    """style_keep / style_jitter are the already-drawn dropout mask [B,1] and jitter [B,Ds] (the
    reference draws them with torch.rand / randn_like inside forward, :104-108)."""
    b = q_src.shape[0]
    if noise is None:
        noise = torch.zeros((b, noise_dim), dtype=q_src.dtype)
    zs = z_style_tgt
    if style_keep is not None:
        zs = zs * style_keep
    if style_jitter is not None:
        zs = zs + style_jitter
    cond = torch.cat([z_content, zs, noise], dim=1)
    n_blocks = len({k.split(".")[1] for k in sd if k.startswith("blocks.")})
    h = F.conv1d(q_src, sd["in_proj.weight"], sd["in_proj.bias"], padding=1)
    for i, d in enumerate(g_dilations(n_blocks)):
        p = f"blocks.{i}."
        r = F.conv1d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=d, dilation=d)
        r = F.group_norm(r, 8, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        r = F.silu(_film(sd, p + "film1.", r, cond))
        r = F.conv1d(r, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
        r = F.group_norm(r, 8, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        r = F.silu(_film(sd, p + "film2.", r, cond))
        h = h + r
    h = F.silu(F.group_norm(h, 8, sd["out_norm.weight"], sd["out_norm.bias"]))
    raw = F.conv1d(h, sd["out_proj.weight"], sd["out_proj.bias"], padding=1)
    if direct_output:
        return raw
    return q_src + residual_scale * torch.tanh(raw)


# ------------------------------------------------------------------------------------------------
# weight norm: w = g * v / ||v|| with the norm over every dim but 0 (torch _weight_norm, dim=0)
def weight_norm_w(g: Tensor, v: Tensor) -> Tensor:
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (v.dim() - 1))
    return v * (g / n)


# ------------------------------------------------------------------------------------------------
# discriminator (lab3_codec_models.py:122-168)
D_BLOCKS = ((1, 32, 15, 1, 7), (32, 64, 41, 4, 20), (64, 128, 41, 4, 20), (128, 256, 41, 4, 20), (256, 512, 41, 4, 20))


def single_scale_d(sd: SD, pre: str, wav: Tensor) -> Tuple[Tensor, List[Tensor]]:
    feats = []
    h = wav
    for j, (_, _, _, s, p) in enumerate(D_BLOCKS):
        w = weight_norm_w(sd[f"{pre}blocks.{j}.conv.weight_g"], sd[f"{pre}blocks.{j}.conv.weight_v"])
        h = F.leaky_relu(F.conv1d(h, w, sd[f"{pre}blocks.{j}.conv.bias"], stride=s, padding=p), 0.2)
        feats.append(h)
    w = weight_norm_w(sd[pre + "out.weight_g"], sd[pre + "out.weight_v"])
    logits = F.conv1d(h, w, sd[pre + "out.bias"], stride=1, padding=1)
    return logits, feats


def discriminator_forward(sd: SD, wav: Tensor) -> List[Tuple[Tensor, List[Tensor]]]:
    n_scales = len({k.split(".")[1] for k in sd if k.startswith("discs.")})
    outs = []
    x = wav
    for i in range(n_scales):
        if i > 0:
            x = F.avg_pool1d(x, kernel_size=4, stride=2, padding=1)  # count_include_pad=True
        outs.append(single_scale_d(sd, f"discs.{i}.", x))
    return outs


# ------------------------------------------------------------------------------------------------
# losses (lab3_codec_models.py:171-215)
def hinge_d_loss(real_logits: Tensor, fake_logits: Tensor) -> Tensor:
    return F.relu(1.0 - real_logits).mean() + F.relu(1.0 + fake_logits).mean()


def hinge_g_loss(fake_logits: Tensor) -> Tensor:
    return -fake_logits.mean()


def feature_matching_loss(real_feats, fake_feats) -> Tensor:
    loss = torch.tensor(0.0)
    for a, b in zip(real_feats, fake_feats):
        loss = loss + (a.detach() - b).abs().mean()
    return loss / max(1, len(fake_feats))


def multiscale_hinge_d_loss(real_outs, fake_outs) -> Tensor:
    loss = torch.tensor(0.0)
    for (rl, _), (fl, _) in zip(real_outs, fake_outs):
        loss = loss + hinge_d_loss(rl, fl)
    return loss / max(1, len(real_outs))


def multiscale_hinge_g_loss(fake_outs) -> Tensor:
    loss = torch.tensor(0.0)
    for fl, _ in fake_outs:
        loss = loss + hinge_g_loss(fl)
    return loss / max(1, len(fake_outs))


def multiscale_feature_matching_loss(real_outs, fake_outs) -> Tensor:
    loss = torch.tensor(0.0)
    for (_, rf), (_, ff) in zip(real_outs, fake_outs):
        loss = loss + feature_matching_loss(rf, ff)
    return loss / max(1, len(real_outs))


# ------------------------------------------------------------------------------------------------
# MR-STFT (lab3_codec_train.py:148-176) restated with explicit framing + rfft instead of torch.stft
def stft_mag(x: Tensor, n_fft: int, hop: int, win: int) -> Tensor:
    """|STFT| of x [B,S] -> [B, n_fft/2+1, 1+S//hop]; center=True (reflect pad n_fft/2), periodic Hann
    of length `win` zero-padded (centered) to n_fft, onesided, unnormalized."""
    n = torch.arange(win, dtype=torch.float32)
    window = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / win)  # torch.hann_window(periodic=True)
    if win < n_fft:
        left = (n_fft - win) // 2
        window = F.pad(window, (left, n_fft - win - left))
    xp = F.pad(x.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop)  # [B, n_frames, n_fft]
    spec = torch.fft.rfft(frames * window, dim=-1)  # [B, n_frames, bins]
    return spec.abs().transpose(1, 2)


def spectral_terms(x: Tensor, y: Tensor, n_fft: int, hop: int, win: int) -> Tuple[Tensor, Tensor]:
    mx = stft_mag(x, n_fft, hop, win).clamp_min(1e-8)
    my = stft_mag(y, n_fft, hop, win).clamp_min(1e-8)
    sc = torch.sqrt(((my - mx) ** 2).sum()) / (torch.sqrt((my ** 2).sum()) + 1e-8)
    lm = (torch.log(mx) - torch.log(my)).abs().mean()
    return sc, lm


def waveform_mrstft_loss(wav_pred: Tensor, wav_true: Tensor, resolutions=DEFAULT_RESOLUTIONS) -> Tensor:
    if wav_pred.shape != wav_true.shape:
        raise ValueError(f"Shape mismatch: {tuple(wav_pred.shape)} vs {tuple(wav_true.shape)}")
    x = wav_pred.squeeze(1)
    y = wav_true.squeeze(1)
    total = torch.tensor(0.0)
    used = 0
    for n_fft, hop, win in resolutions:
        if x.shape[-1] < int(n_fft):
            continue
        sc, lm = spectral_terms(x, y, int(n_fft), int(hop), int(win))
        total = total + sc + lm
        used += 1
    if used == 0:
        return (wav_pred - wav_true).abs().mean()
    return total / float(used)


def mel_mrstft_loss(mel_pred: Tensor, mel_true: Tensor, resolutions=((64, 16, 64), (128, 32, 128), (256, 64, 256))) -> Tensor:
    """lab 3/src/lab3_losses.py:16-71 (multi_resolution_stft_loss): the waveform loss's sibling on mel trajectories
    [B, n_mels, T], magnitudes |X| + 1e-8; resolutions with n_fft > T are skipped; L1 fallback when none is used."""
    if mel_pred.shape != mel_true.shape:
        raise ValueError(f"Shape mismatch: {mel_pred.shape} vs {mel_true.shape}")
    b, n_mels, t = mel_pred.shape
    x, y = mel_pred.reshape(b * n_mels, t), mel_true.reshape(b * n_mels, t)
    total, count = 0.0, 0
    for n_fft, hop, win in resolutions:
        if t < int(n_fft):
            continue
        window = torch.hann_window(int(win), dtype=x.dtype)
        sx = torch.stft(x, n_fft=int(n_fft), hop_length=int(hop), win_length=int(win), window=window, return_complex=True, center=True)
        sy = torch.stft(y, n_fft=int(n_fft), hop_length=int(hop), win_length=int(win), window=window, return_complex=True, center=True)
        mx, my = sx.abs() + 1e-8, sy.abs() + 1e-8
        total = total + torch.norm(my - mx, p="fro") / (torch.norm(my, p="fro") + 1e-8) + F.l1_loss(torch.log(mx), torch.log(my))
        count += 1
    if count == 0:
        return F.l1_loss(mel_pred, mel_true)
    return total / float(count)


# ------------------------------------------------------------------------------------------------
# frozen SEANet decoder (HF modeling_encodec.py, EncodecConfig() defaults = 24 kHz model)
RATIOS = (8, 5, 4, 2)


def _wn(sd: SD, pre: str) -> Tensor:
    return weight_norm_w(sd[pre + "parametrizations.weight.original0"], sd[pre + "parametrizations.weight.original1"])


def _causal_conv(sd: SD, pre: str, x: Tensor, dilation: int = 1) -> Tensor:
    """EncodecConv1d.forward (:157-176): causal, stride 1 => extra padding 0; reflect left pad k_eff-1.
    _pad1d (:139-155) zero-extends on the right first when the input is not longer than the pad."""
    w = _wn(sd, pre)
    k_eff = (w.shape[-1] - 1) * dilation + 1
    pad = k_eff - 1
    if pad > 0:
        length = x.shape[-1]
        extra = 0
        if length <= pad:
            extra = pad - length + 1
            x = F.pad(x, (0, extra))
        x = F.pad(x, (pad, 0), mode="reflect")
        if extra:
            x = x[..., : x.shape[-1] - extra]
    return F.conv1d(x, w, sd[pre + "bias"], dilation=dilation)


def _causal_convtr(sd: SD, pre: str, x: Tensor, stride: int) -> Tensor:
    """EncodecConvTranspose1d.forward (:206-233): trim_right_ratio=1 => drop k-stride samples on the right."""
    w = _wn(sd, pre)  # [Cin, Cout, k], norm per input channel
    y = F.conv_transpose1d(x, w, sd[pre + "bias"], stride=stride)
    trim = w.shape[-1] - stride
    return y[..., : y.shape[-1] - trim]


def _lstm(sd: SD, pre: str, x: Tensor) -> Tensor:
    """EncodecLSTM.forward (:244-249): 2-layer LSTM over time + skip.  Gate order i,f,g,o (torch nn.LSTM)."""
    seq = x.permute(2, 0, 1)  # [T,B,C]
    inp = seq
    for layer in range(2):
        w_ih, w_hh = sd[f"{pre}lstm.weight_ih_l{layer}"], sd[f"{pre}lstm.weight_hh_l{layer}"]
        bias = sd[f"{pre}lstm.bias_ih_l{layer}"] + sd[f"{pre}lstm.bias_hh_l{layer}"]
        hdim = w_hh.shape[1]
        xg = F.linear(inp, w_ih, bias)  # [T,B,4H]
        h = torch.zeros(inp.shape[1], hdim)
        c = torch.zeros(inp.shape[1], hdim)
        outs = []
        for t in range(inp.shape[0]):
            g = xg[t] + F.linear(h, w_hh)
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, dim=0)
    return (inp + seq).permute(1, 2, 0)


def _lstm_fast(sd: SD, pre: str, x: Tensor) -> Tensor:
    """Same arithmetic through torch's stock fused CPU LSTM op (used for the timed CPU baseline)."""
    seq = x.permute(2, 0, 1)
    flat = []
    for layer in range(2):
        flat += [sd[f"{pre}lstm.weight_ih_l{layer}"], sd[f"{pre}lstm.weight_hh_l{layer}"],
                 sd[f"{pre}lstm.bias_ih_l{layer}"], sd[f"{pre}lstm.bias_hh_l{layer}"]]
    hdim = flat[1].shape[1]
    z = torch.zeros(2, seq.shape[1], hdim)
    out, _, _ = torch._VF.lstm(seq, (z, z), flat, True, 2, 0.0, False, False, False)
    return (out + seq).permute(1, 2, 0)


def decoder_forward(sd: SD, q: Tensor, fast_lstm: bool = True, collect: Optional[list] = None) -> Tensor:
    """EncodecDecoder.forward (:343-347) with the default layer list (:320-341):
    0 conv k7 | 1 lstm | then per ratio: ELU, convT(k=2r,s=r), resblock | ELU | conv k7."""
    def rec(h):
        if collect is not None:
            collect.append(h)
        return h
    h = rec(_causal_conv(sd, "layers.0.conv.", q))
    h = rec((_lstm_fast if fast_lstm else _lstm)(sd, "layers.1.", h))
    idx = 2
    for r in RATIOS:
        h = rec(F.elu(h)); idx += 1
        h = rec(_causal_convtr(sd, f"layers.{idx}.conv.", h, r)); idx += 1
        # EncodecResnetBlock (:252-283): ELU, conv k3 (C->C/2), ELU, conv k1 (C/2->C) + conv-shortcut k1
        p = f"layers.{idx}."
        t = _causal_conv(sd, p + "block.1.conv.", F.elu(h))
        t = _causal_conv(sd, p + "block.3.conv.", F.elu(t))
        h = rec(_causal_conv(sd, p + "shortcut.conv.", h) + t); idx += 1
    h = rec(F.elu(h)); idx += 1
    return rec(_causal_conv(sd, f"layers.{idx}.conv.", h))


# ------------------------------------------------------------------------------------------------
# CodecStyleJudge (lab3_codec_judge.py:83-145) and the codec_judge_ce style terms (lab3_codec_train.py:480-494)
def style_judge_embed(sd: SD, q_emb: Tensor) -> Tensor:
    """net = 3 x [Conv1d -> GroupNorm(8) -> SiLU] (k5 p2, k3 p1, k3 p1), mean over time, proj, L2-normalise (:131-135)."""
    h = q_emb
    for ci, gi, pad in ((0, 1, 2), (3, 4, 1), (6, 7, 1)):
        h = F.conv1d(h, sd[f"net.{ci}.weight"], sd[f"net.{ci}.bias"], padding=pad)
        h = F.silu(F.group_norm(h, 8, sd[f"net.{gi}.weight"], sd[f"net.{gi}.bias"], eps=1e-5))
    e = F.linear(h.mean(dim=-1), sd["proj.weight"], sd["proj.bias"])
    return F.normalize(e, dim=-1)


def style_judge_terms(sd: SD, q_hat: Tensor, z_style_tgt: Tensor, tgt_genre_idx: Tensor, src_genre_idx: Tensor,
                      embed_align_weight: float, push_margin: float, stage1: bool = False):
    """(loss_style, loss_style_embed, loss_style_push, emb, logits) as the reference's G step forms them (:480-494)."""
    emb = style_judge_embed(sd, q_hat)
    logits = F.linear(emb, sd["head.weight"], sd["head.bias"])
    loss_style = F.cross_entropy(logits, tgt_genre_idx)
    loss_embed = torch.zeros(())
    if embed_align_weight > 0.0 and z_style_tgt.shape[1] == emb.shape[1]:
        loss_embed = (1.0 - F.cosine_similarity(emb, F.normalize(z_style_tgt, dim=-1), dim=-1)).mean()
        loss_style = loss_style + embed_align_weight * loss_embed
    loss_push = torch.zeros(())
    if not stage1:
        p_src = torch.softmax(logits, dim=1).gather(1, src_genre_idx.view(-1, 1)).squeeze(1)
        loss_push = F.relu(p_src - push_margin).mean()
    return loss_style, loss_embed, loss_push, emb, logits


# ------------------------------------------------------------------------------------------------
# clip_grad_norm_ + AdamW (torch defaults: weight_decay 0.01, eps 1e-8, no amsgrad)
@dataclass
class AdamWState:
    lr: float = 2e-4
    betas: Tuple[float, float] = (0.5, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.01
    step: Dict[str, int] = field(default_factory=dict)
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> float:
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = min(1.0, float(max_norm) / (float(total) + 1e-6))
    if coef < 1.0:
        for g in grads.values():
            g.mul_(coef)
    return float(total)


@torch.no_grad()
def adamw_step(params: SD, grads: Dict[str, Tensor], st: AdamWState) -> None:
    b1, b2 = st.betas
    for k, g in grads.items():
        p = params[k]
        if k not in st.m:
            st.m[k] = torch.zeros_like(p); st.v[k] = torch.zeros_like(p); st.step[k] = 0
        st.step[k] += 1
        t = st.step[k]
        p.mul_(1.0 - st.lr * st.weight_decay)
        st.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
        st.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        denom = (st.v[k].sqrt() / math.sqrt(bc2)).add_(st.eps)
        p.addcdiv_(st.m[k], denom, value=-(st.lr / bc1))


# ------------------------------------------------------------------------------------------------
@dataclass
class Weights:  # CodecTrainWeights (lab3_codec_models.py:218-229)
    adv: float = 0.5
    latent_l1: float = 4.0
    latent_continuity: float = 1.0
    content: float = 2.0
    style: float = 3.0
    mrstft: float = 2.0
    feature_match: float = 1.0
    mode_seeking: float = 0.0
    style_push: float = 0.0
    delta_budget: float = 0.0


@dataclass
class StepCfg:
    d_step_period: int = 1
    r1_gamma: float = 0.0
    r1_interval: int = 16
    g_grad_clip_norm: float = 5.0
    d_grad_clip_norm: float = 5.0
    delta_budget: float = 0.12
    resolutions: Sequence[Tuple[int, int, int]] = DEFAULT_RESOLUTIONS
    weights: Weights = field(default_factory=Weights)


def _leafify(sd: SD) -> SD:
    return {k: v.detach().requires_grad_(True) for k, v in sd.items()}


def iteration(g_sd: SD, d_sd: SD, dec_sd: SD, opt_g: AdamWState, opt_d: AdamWState, batch, global_step: int,
              cfg: StepCfg, fast_lstm: bool = True, record: bool = False) -> dict:
    """One iteration of train_codec_stage's loop body (lab3_codec_train.py:404-591), north-star subset.
    g_sd / d_sd are updated in place.  batch = (q_src, zc, z_style_tgt, noise, q_real_d, q_real_g)."""
    q_src, zc, zs_tgt, noise, q_real_d, q_real_g = batch
    out = {}
    gp = _leafify(g_sd)
    q_hat = generator_forward(gp, q_src, zc, zs_tgt, noise)
    x_hat = decoder_forward(dec_sd, q_hat, fast_lstm)
    with torch.no_grad():
        x_src = decoder_forward(dec_sd, q_src, fast_lstm)
    take_d = (global_step % max(1, int(cfg.d_step_period))) == 0
    out["take_d_step"] = float(take_d)
    if take_d:
        dp = _leafify(d_sd)
        with torch.no_grad():
            x_real = decoder_forward(dec_sd, q_real_d, fast_lstm)
        x_real = x_real.detach().requires_grad_(cfg.r1_gamma > 0.0)
        real_outs = discriminator_forward(dp, x_real)
        fake_outs_d = discriminator_forward(dp, x_hat.detach())
        loss_d = multiscale_hinge_d_loss(real_outs, fake_outs_d)
        out["loss_d_hinge"] = float(loss_d.detach())
        do_r1 = cfg.r1_gamma > 0.0 and (global_step % max(1, int(cfg.r1_interval)) == 0)
        out["do_r1"] = float(do_r1)
        if do_r1:
            real_sum = sum(rl.mean() for rl, _ in real_outs)
            grad, = torch.autograd.grad(real_sum, x_real, create_graph=True, retain_graph=True)
            r1 = grad.reshape(grad.shape[0], -1).pow(2).sum(dim=1).mean()
            out["r1"] = float(r1.detach())
            loss_d = loss_d + 0.5 * cfg.r1_gamma * r1
        names = list(dp.keys())
        gl = torch.autograd.grad(loss_d, [dp[k] for k in names])
        grads = {k: g.clone() for k, g in zip(names, gl)}
        if record:
            out["d_grads"] = {k: g.clone() for k, g in grads.items()}
        if cfg.d_grad_clip_norm > 0.0:
            out["d_grad_norm"] = clip_grad_norm(grads, cfg.d_grad_clip_norm)
        adamw_step(d_sd, grads, opt_d)
        out["loss_d"] = float(loss_d.detach())

    with torch.no_grad():
        x_real_g = decoder_forward(dec_sd, q_real_g, fast_lstm)
    dconst = {k: v.detach() for k, v in d_sd.items()}  # D grads from the G step are discarded by the reference
    real_outs_g = discriminator_forward(dconst, x_real_g)
    fake_outs = discriminator_forward(dconst, x_hat)
    loss_adv = multiscale_hinge_g_loss(fake_outs)
    loss_fm = multiscale_feature_matching_loss(real_outs_g, fake_outs)
    loss_lat_l1 = (q_hat - q_src).abs().mean()
    loss_lat_cont = ((q_hat[:, :, 1:] - q_hat[:, :, :-1]) - (q_src[:, :, 1:] - q_src[:, :, :-1])).abs().mean()
    loss_mrstft = waveform_mrstft_loss(x_hat, x_src, cfg.resolutions)
    delta_mean = (q_hat - q_src).abs().mean(dim=(1, 2))
    loss_db = F.relu(delta_mean - cfg.delta_budget).mean()
    w = cfg.weights
    loss_g = (w.adv * loss_adv + w.feature_match * loss_fm + w.latent_l1 * loss_lat_l1
              + w.latent_continuity * loss_lat_cont + w.mrstft * loss_mrstft + w.delta_budget * loss_db)
    names = list(gp.keys())
    gl = torch.autograd.grad(loss_g, [gp[k] for k in names])
    grads = {k: g.clone() for k, g in zip(names, gl)}
    if record:
        out["g_grads"] = {k: g.clone() for k, g in grads.items()}
        out["q_hat"] = q_hat.detach(); out["x_hat"] = x_hat.detach(); out["x_src"] = x_src.detach()
    if cfg.g_grad_clip_norm > 0.0:
        out["g_grad_norm"] = clip_grad_norm(grads, cfg.g_grad_clip_norm)
    adamw_step(g_sd, grads, opt_g)
    out.update(loss_g=float(loss_g.detach()), loss_adv=float(loss_adv.detach()), loss_fm=float(loss_fm.detach()),
               loss_lat_l1=float(loss_lat_l1.detach()), loss_lat_cont=float(loss_lat_cont.detach()),
               loss_mrstft=float(loss_mrstft.detach()), loss_delta_budget=float(loss_db.detach()))
    return out


# ------------------------------------------------------------------------------------------------
# host-side helpers on the path (lab3_codec_train.py:59-145)
def build_style_centroid_bank(z_style, genre_idx, n_genres):
    import numpy as np
    bank = np.zeros((int(n_genres), z_style.shape[1]), dtype=np.float32)
    for g in range(int(n_genres)):
        mask = genre_idx == g
        if np.any(mask):
            v = z_style[mask].mean(axis=0).astype(np.float32)
            bank[g] = v / (np.linalg.norm(v) + 1e-8)
    return torch.from_numpy(bank)


def shift_targets_balanced(source_idx: Tensor, n_genres: int, offset: int = 0) -> Tensor:
    n = source_idx.shape[0]
    tgt = (torch.arange(n, dtype=torch.long) + int(offset)) % int(n_genres)
    clash = tgt == source_idx
    tgt[clash] = (tgt[clash] + 1) % int(n_genres)
    return tgt


# ------------------------------------------------------------------------------------------------
def fresh_state(seed_g: int = 1, seed_d: int = 2, seed_dec: int = 3, z_style_dim: int = 128, hidden: int = 256,
                n_blocks: int = 10, n_scales: int = 3):
    """Procedurally filled (g_sd, d_sd, dec_sd) with the reference's key names and shapes."""
    from .procfill import fill_state_dict
    from .shapes import decoder_shapes, discriminator_shapes, generator_shapes
    mk = lambda shapes, seed: fill_state_dict({k: torch.empty(s) for k, s in shapes.items()}, seed)
    return (mk(generator_shapes(z_style_dim=z_style_dim, hidden=hidden, n_blocks=n_blocks), seed_g),
            mk(discriminator_shapes(n_scales), seed_d), mk(decoder_shapes(), seed_dec))
