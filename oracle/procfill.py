"""Procedural (closed-form) tensor fill shared by the golden-vector generator and the tests.

TEST INFRASTRUCTURE ONLY.  Nothing under gan_amd/ may import this module.

Every value is an exact integer hash of (flat index, key hash, seed) mapped to [-1, 1), so the
reference modules (in the build container) and this repo's modules (anywhere) can be filled with
bit-identical parameters without committing weight files.  The reference ships no weights for this
path (lab3 checkpoints are git-ignored; the EnCodec weights are a remote fetch).
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in [-1, 1), pure uint64 arithmetic (portable, no libm)."""
    idx = np.arange(n, dtype=np.uint64)
    s64 = np.uint64((int(seed) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        h = idx * np.uint64(0x9E3779B97F4A7C15) + s64
        return _mix(h)


def _mix(h: np.ndarray) -> np.ndarray:
    h ^= h >> np.uint64(30)
    h = h * np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(27)
    h = h * np.uint64(0x94D049BB133111EB)
    h ^= h >> np.uint64(31)
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 / float(1 << 53)) - 1.0


def key_seed(key: str, seed: int) -> int:
    return (zlib.crc32(key.encode("utf-8")) * 1000003 + int(seed)) & 0x7FFFFFFF


def proc_tensor(shape, key: str, seed: int, scale: float = 1.0, offset: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(n, key_seed(key, seed))
    return torch.from_numpy((offset + scale * u).astype(np.float32).reshape(tuple(shape)))


def _rule(key: str, shape) -> tuple:
    """(scale, offset) for a parameter, chosen so activations stay O(1) through deep stacks."""
    nd = len(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("weight_g", "original0"):
        return 0.25, 1.25  # weight-norm gain, strictly positive
    if leaf in ("weight_v", "original1"):
        return 1.0, 0.0  # direction only; row norm divides out
    if "lstm" in key:
        return (1.0 / np.sqrt(512.0)), 0.0
    if leaf == "running_var":  # BatchNorm buffer, strictly positive
        return 0.25, 1.0
    if nd == 4:  # Conv2d [Cout, Cin, kh, kw] / ConvTranspose2d [Cin, Cout, kh, kw] weight (the mel track)
        return float(np.sqrt(3.0 / (shape[1] * shape[2] * shape[3]))), 0.0
    if nd == 3:  # plain Conv1d weight [Cout, Cin, k]
        fan_in = shape[1] * shape[2]
        return float(np.sqrt(3.0 / fan_in)), 0.0
    if nd == 2:  # Linear weight [out, in]
        return float(np.sqrt(3.0 / shape[1])), 0.0
    if nd == 1 and leaf == "weight":  # GroupNorm gamma
        return 0.1, 1.0
    return 0.05, 0.0  # biases / GroupNorm beta


def fill_state_dict(sd: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """Return a new state dict with the same keys/shapes/dtypes, procedurally filled."""
    out = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        s, o = _rule(k, tuple(v.shape))
        out[k] = proc_tensor(tuple(v.shape), k, seed, s, o).to(v.dtype)
    return out


def proc_input(shape, name: str, seed: int, scale: float = 1.0) -> torch.Tensor:
    """Procedural 'data' tensor (inputs, noise, bank rows)."""
    return proc_tensor(shape, "input:" + name, seed, scale, 0.0)
