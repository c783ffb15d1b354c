"""State-dict shapes of the reference's mel-track modules (lab 3/src/lab3_models.py), written out so that tests can build
procedurally filled state dicts without the reference (which does not travel to the GPU box)."""
from typing import Dict, Tuple


def decoder_state_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """ReconstructionDecoder.state_dict() (lab3_models.py:107-114) for a DecoderCfg-like object."""
    b, h0, w0 = cfg.base_channels, cfg.n_mels // 8, cfg.n_frames // 8
    s = {"fc.weight": (b * h0 * w0, cfg.zc_dim + cfg.cond_dim), "fc.bias": (b * h0 * w0,)}
    chans = [(b, b // 2), (b // 2, b // 4), (b // 4, b // 8)]
    for i, (ci, co) in enumerate(chans, start=1):
        if cfg.upsample == "transpose":
            s[f"up{i}.0.weight"], s[f"up{i}.0.bias"], ni = (ci, co, 4, 4), (co,), 1
        elif cfg.upsample == "nearest":
            s[f"up{i}.1.weight"], s[f"up{i}.1.bias"], ni = (co, ci, 3, 3), (co,), 2
        else:
            s[f"up{i}.0.weight"], s[f"up{i}.0.bias"], ni = (co * 4, ci, 3, 3), (co * 4,), 2
        s[f"up{i}.{ni}.weight"], s[f"up{i}.{ni}.bias"] = (co,), (co,)
        if cfg.norm == "batch":
            s[f"up{i}.{ni}.running_mean"], s[f"up{i}.{ni}.running_var"], s[f"up{i}.{ni}.num_batches_tracked"] = (co,), (co,), ()
        if cfg.mrf:
            for j, k in enumerate(cfg.mrf_kernels):
                s[f"mrf{i}.branches.{j}.0.weight"], s[f"mrf{i}.branches.{j}.0.bias"] = (co, co, int(k), int(k)), (co,)
                s[f"mrf{i}.branches.{j}.2.weight"], s[f"mrf{i}.branches.{j}.2.bias"] = (co, co, 3, 3), (co,)
            s[f"mrf{i}.out.weight"], s[f"mrf{i}.out.bias"] = (co, co, 1, 1), (co,)
    s["out.weight"], s["out.bias"] = (1, b // 8, 3, 3), (1,)
    return _module_order(s, cfg)


def _module_order(s, cfg):
    """nn.Module registration order: fc, up1, up2, up3, mrf1, mrf2, mrf3, out (lab3_models.py:107-114)."""
    order = ["fc."] + [f"up{i}." for i in (1, 2, 3)] + [f"mrf{i}." for i in (1, 2, 3)] + ["out."]
    return {k: v for pre in order for k, v in s.items() if k.startswith(pre)}


def disc_state_shapes(base: int = 64, cond_dim: int = 0, pre: str = "") -> Dict[str, Tuple[int, ...]]:
    """MelDiscriminator.state_dict() (lab3_models.py:155-174)."""
    c = base
    s = {pre + "net.0.weight": (c, 1, 4, 4), pre + "net.0.bias": (c,)}
    for ci, bi, cin, cout in ((2, 3, c, 2 * c), (5, 6, 2 * c, 4 * c), (8, 9, 4 * c, 8 * c)):
        s[f"{pre}net.{ci}.weight"], s[f"{pre}net.{ci}.bias"] = (cout, cin, 4, 4), (cout,)
        s[f"{pre}net.{bi}.weight"], s[f"{pre}net.{bi}.bias"] = (cout,), (cout,)
        s[f"{pre}net.{bi}.running_mean"], s[f"{pre}net.{bi}.running_var"], s[f"{pre}net.{bi}.num_batches_tracked"] = (cout,), (cout,), ()
    s[pre + "head.weight"], s[pre + "head.bias"] = (1, 8 * c), (1,)
    if cond_dim > 0:
        s[pre + "cond_proj.weight"], s[pre + "cond_proj.bias"] = (8 * c, cond_dim), (8 * c,)
    return s


def wrapper_state_shapes(kind: str, base: int = 64, cond_dim: int = 0) -> Dict[str, Tuple[int, ...]]:
    """The four wrappers (lab3_models.py:194-408): prefixes of their sub-discriminators."""
    if kind == "single":
        return disc_state_shapes(base, cond_dim)
    s = {}
    if kind == "multiscale":
        for i in range(3):
            s.update(disc_state_shapes(base, cond_dim, f"discriminators.{i}."))
    elif kind == "subband":
        for n in ("d_low", "d_mid", "d_high"):
            s.update(disc_state_shapes(base, cond_dim, f"{n}."))
    elif kind == "multiperiod":
        for i in range(4):
            s.update(disc_state_shapes(base, cond_dim, f"discriminators.{i}."))
    elif kind == "hybrid":
        for i in range(3):
            s.update(disc_state_shapes(64, cond_dim, f"ms.discriminators.{i}."))
        for i in range(4):
            s.update(disc_state_shapes(64, cond_dim, f"mp.discriminators.{i}."))
    else:
        raise ValueError(kind)
    return s
