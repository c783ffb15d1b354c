"""State-dict key -> shape tables for G, D and the SEANet decoder (reference key names).

TEST INFRASTRUCTURE ONLY.  Key names follow lab3_codec_models.py (nn.Module attribute names, old-style
nn.utils.weight_norm => weight_g / weight_v) and HF EncodecDecoder (parametrizations.weight.original0/1).
"""
from __future__ import annotations

from typing import Dict, Tuple


def generator_shapes(in_channels=128, z_content_dim=128, z_style_dim=128, hidden=256, n_blocks=10,
                     noise_dim=32) -> Dict[str, Tuple[int, ...]]:
    cond = z_content_dim + z_style_dim + noise_dim
    s: Dict[str, Tuple[int, ...]] = {"in_proj.weight": (hidden, in_channels, 3), "in_proj.bias": (hidden,)}
    for i in range(n_blocks):
        p = f"blocks.{i}."
        for c in ("conv1", "conv2"):
            s[p + c + ".weight"] = (hidden, hidden, 3)
            s[p + c + ".bias"] = (hidden,)
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (hidden,)
            s[p + n + ".bias"] = (hidden,)
        for f in ("film1", "film2"):
            s[p + f + ".proj.0.weight"] = (2 * hidden, cond)
            s[p + f + ".proj.0.bias"] = (2 * hidden,)
            s[p + f + ".proj.2.weight"] = (2 * hidden, 2 * hidden)
            s[p + f + ".proj.2.bias"] = (2 * hidden,)
    s["out_norm.weight"] = (hidden,)
    s["out_norm.bias"] = (hidden,)
    s["out_proj.weight"] = (in_channels, hidden, 3)
    s["out_proj.bias"] = (in_channels,)
    return s


D_BLOCKS = ((1, 32, 15), (32, 64, 41), (64, 128, 41), (128, 256, 41), (256, 512, 41))


def discriminator_shapes(n_scales=3) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    for i in range(n_scales):
        for j, (ci, co, k) in enumerate(D_BLOCKS):
            p = f"discs.{i}.blocks.{j}.conv."
            s[p + "bias"] = (co,)
            s[p + "weight_g"] = (co, 1, 1)
            s[p + "weight_v"] = (co, ci, k)
        p = f"discs.{i}.out."
        s[p + "bias"] = (1,)
        s[p + "weight_g"] = (1, 1, 1)
        s[p + "weight_v"] = (1, 512, 3)
    return s


def decoder_shapes(hidden_size=128, num_filters=32, ratios=(8, 5, 4, 2), kernel_size=7, last_kernel_size=7,
                   residual_kernel_size=3, compress=2, lstm_layers=2) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(pre, cout, cin, k):
        s[pre + "conv.bias"] = (cout,)
        s[pre + "conv.parametrizations.weight.original0"] = (cout, 1, 1)
        s[pre + "conv.parametrizations.weight.original1"] = (cout, cin, k)

    scaling = 2 ** len(ratios)
    c = scaling * num_filters
    conv("layers.0.", c, hidden_size, kernel_size)
    for l in range(lstm_layers):
        s[f"layers.1.lstm.weight_ih_l{l}"] = (4 * c, c)
        s[f"layers.1.lstm.weight_hh_l{l}"] = (4 * c, c)
        s[f"layers.1.lstm.bias_ih_l{l}"] = (4 * c,)
        s[f"layers.1.lstm.bias_hh_l{l}"] = (4 * c,)
    idx = 2
    for r in ratios:
        idx += 1  # ELU
        pre = f"layers.{idx}."
        s[pre + "conv.bias"] = (c // 2,)
        s[pre + "conv.parametrizations.weight.original0"] = (c, 1, 1)  # ConvTranspose1d: g per input channel
        s[pre + "conv.parametrizations.weight.original1"] = (c, c // 2, 2 * r)
        idx += 1
        c //= 2
        pre = f"layers.{idx}."
        conv(pre + "block.1.", c // compress, c, residual_kernel_size)
        conv(pre + "block.3.", c, c // compress, 1)
        conv(pre + "shortcut.", c, c, 1)
        idx += 1
    idx += 1  # ELU
    conv(f"layers.{idx}.", 1, num_filters, last_kernel_size)
    return s


def style_judge_shapes(in_channels=128, n_genres=10, hidden=256, emb_dim=128) -> Dict[str, Tuple[int, ...]]:
    """CodecStyleJudge without a source head (lab3_codec_judge.py:92-113): net.{0,3,6} Conv1d, net.{1,4,7} GroupNorm."""
    s: Dict[str, Tuple[int, ...]] = {}
    for slot, (ci, k) in zip((0, 3, 6), ((in_channels, 5), (hidden, 3), (hidden, 3))):
        s[f"net.{slot}.weight"] = (hidden, ci, k)
        s[f"net.{slot}.bias"] = (hidden,)
        s[f"net.{slot + 1}.weight"] = (hidden,)
        s[f"net.{slot + 1}.bias"] = (hidden,)
    s["proj.weight"], s["proj.bias"] = (emb_dim, hidden), (emb_dim,)
    s["head.weight"], s["head.bias"] = (n_genres, emb_dim), (n_genres,)
    return s
