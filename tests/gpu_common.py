"""Helpers shared by the GPU parity tests, smoke() and bench.py."""
from __future__ import annotations

import torch

from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.lab3_codec_train import CodecStageTrainConfig, codec_train_iteration
from gan_amd.optim import FusedAdamW


def build_gpu_models(g_sd, d_sd, dec_sd, precision: str, device="cuda", **gkw):
    G = CodecLatentTranslator(precision=precision, **gkw)
    G.load_state_dict(g_sd, strict=True)
    n_scales = len({k.split(".")[1] for k in d_sd})
    D = MultiScaleWaveDiscriminator(n_scales, precision=precision)
    D.load_state_dict(d_sd, strict=True)
    codec = FrozenEncodec(decoder_state_dict=dec_sd, random_init=dec_sd is None, device=device, precision=precision)
    return G.to(device).train(), D.to(device).train(), codec


def stage_cfg_from_oracle(cfg) -> CodecStageTrainConfig:
    w = cfg.weights
    return CodecStageTrainConfig(
        stage_name="stage2", epochs=1, d_step_period=cfg.d_step_period, r1_gamma=cfg.r1_gamma, r1_interval=cfg.r1_interval,
        g_grad_clip_norm=cfg.g_grad_clip_norm, d_grad_clip_norm=cfg.d_grad_clip_norm, delta_budget=cfg.delta_budget,
        wave_mrstft_resolutions=tuple(cfg.resolutions),
        weights=CodecTrainWeights(adv=w.adv, latent_l1=w.latent_l1, latent_continuity=w.latent_continuity, content=0.0,
                                  style=0.0, mrstft=w.mrstft, feature_match=w.feature_match, mode_seeking=w.mode_seeking,
                                  style_push=0.0, delta_budget=w.delta_budget))


def run_gpu_iteration(G, D, codec, batch, cfg, global_step, opt_g=None, opt_d=None, device="cuda"):
    scfg = stage_cfg_from_oracle(cfg)
    opt_g = opt_g or FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    opt_d = opt_d or FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    q_src, zc, zs, noise, q_real_d, q_real_g = [t.to(device) for t in batch]
    out = codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, q_src, zc, zs, noise, q_real_d, q_real_g, global_step)
    out["opt_g"], out["opt_d"] = opt_g, opt_d
    return out


def grab_flat_grads(opts: dict) -> dict:
    """Wrap each FusedAdamW.step so the flat gradient buffer, as the optimizer sees it right before its update (after the
    all-reduce, before the clip), is kept: returns the dict the copies land in."""
    grabbed = {}
    for name, opt in opts.items():
        orig = opt.step

        def step(grads_gathered=False, _o=orig, _opt=opt, _n=name):
            grabbed[_n] = _opt.flat_g.detach().clone()
            return _o(grads_gathered=grads_gathered)
        opt.step = step
    return grabbed


def flat_in_named_order(mod, opt, flat: torch.Tensor) -> torch.Tensor:
    """The optimizer's flat buffer re-ordered as torch.cat over mod.named_parameters() (the order the fixtures use)."""
    by_id = {id(p): (off, p.numel()) for p, off in zip(opt.params, opt.offsets)}
    return torch.cat([flat[by_id[id(p)][0]:by_id[id(p)][0] + by_id[id(p)][1]] for _, p in mod.named_parameters() if p.requires_grad])
