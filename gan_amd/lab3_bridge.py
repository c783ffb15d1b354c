"""The frozen Lab1 encoder of the reference's content / style terms (lab 3/src/lab3_bridge.py) with its convolutions on the HIP kernels.

SURVEY.md section 8(f) next-1 ("... -> dB -> 3 x Conv2d encoder"): `ChunkEncoder` (:34-76) and `FrozenLab1Encoder` (:159-234) with the
reference's names, constructor arguments, attributes, output dictionary and state-dict keys, so a Lab1 checkpoint's `model` entry loads
unchanged.  Both training tracks call `forward_log_mel_tensor` inside the generator step (lab3_codec_train.py:472-550,
lab3_train.py:432-436, :808-916); as a plain torch module on this stack its one-input-channel Conv2d falls to a per-sample im2col
loop (170 launches and 3.7 ms per codec step at B=64).  Here the three stride-2 convolutions run as 1-D implicit GEMMs behind the
kernel-row gather (gan_amd/conv2d.py); BatchNorm (running statistics when frozen), ReLU, the pooled mean and the Linear heads are the
torch modules that hold the parameters.  Also here: the pure tensor / array helpers of that file (`normalize_log_mel`,
`denormalize_log_mel`, `fix_log_mel_frames`, `descriptor32_from_logmel`, `grad_reverse`).  Audio decoding (`load_audio_chunk`,
`extract_log_mel`, `infer_file_with_mel`: librosa) is the reference's I/O and stays there.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .lab3_models import _dtype, _run

MEL_DB_MIN, MEL_DB_MAX = -80.0, 0.0


class _GradientReversal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = float(lambda_)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return -ctx.lambda_ * grad_output, None


def grad_reverse(x: torch.Tensor, lambda_: float = 1.0) -> torch.Tensor:
    """reference lab3_bridge.py:19-31."""
    return _GradientReversal.apply(x, lambda_)


class ChunkEncoder(nn.Module):
    """reference lab3_bridge.py:34-76: log-mel [B, 96, T] -> shared features and the L2-normalised content / style latents.
    Extra keyword: precision="fp32"|"bf16" for the convolutions' operands (parameters and accumulation stay fp32)."""

    def __init__(self, n_sources: int, z_dim: int = 128, precision: str = "fp32"):
        super().__init__()
        self.backbone = nn.Sequential(
            nn.Conv2d(1, 32, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(32), nn.ReLU(),
            nn.Conv2d(32, 64, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(64), nn.ReLU(),
            nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(128), nn.ReLU(),
            nn.AdaptiveAvgPool2d((1, 1)))
        self.shared = nn.Linear(128, 256)
        self.content_head = nn.Linear(256, z_dim)
        self.style_head = nn.Linear(256, z_dim)
        self.style_cls = nn.Linear(z_dim, n_sources)
        self.content_style_adv = nn.Sequential(nn.Linear(z_dim, z_dim), nn.ReLU(), nn.Linear(z_dim, n_sources))
        self.music_head = nn.Linear(256, 1)
        self.precision = precision
        self._dt = _dtype(precision)

    def forward(self, log_mel: torch.Tensor, grl_lambda: float = 1.0) -> Dict[str, torch.Tensor]:
        L.require_cuda(log_mel)
        x = log_mel.unsqueeze(1).to(self._dt).contiguous(memory_format=torch.channels_last)
        h = _run(list(self.backbone), x).flatten(1).float()
        h = F.relu(self.shared(h))
        z_content = F.normalize(self.content_head(h), dim=-1)
        z_style = F.normalize(self.style_head(h), dim=-1)
        return {"shared_feat": h, "z_content": z_content, "z_style": z_style, "style_logits": self.style_cls(z_style),
                "content_style_logits": self.content_style_adv(grad_reverse(z_content, lambda_=grl_lambda)),
                "music_logit": self.music_head(h).squeeze(-1)}


def normalize_log_mel(log_mel_db: np.ndarray) -> np.ndarray:
    """reference :128-132: dB in [-80, 0] -> [-1, 1]."""
    x = (np.clip(log_mel_db, MEL_DB_MIN, MEL_DB_MAX) - MEL_DB_MIN) / (MEL_DB_MAX - MEL_DB_MIN)
    return (x * 2.0 - 1.0).astype(np.float32)


def denormalize_log_mel(mel_norm: torch.Tensor) -> torch.Tensor:
    """reference :135-139: [-1, 1] -> dB."""
    return (torch.clamp(mel_norm, -1.0, 1.0) + 1.0) * (0.5 * (MEL_DB_MAX - MEL_DB_MIN)) + MEL_DB_MIN


def fix_log_mel_frames(log_mel: np.ndarray, n_frames: int) -> np.ndarray:
    """reference :114-125: centre crop, or right-pad with -80 dB, to n_frames."""
    if log_mel.ndim != 2:
        raise ValueError(f"Expected [n_mels, n_frames], got {log_mel.shape}")
    n_mels, t = log_mel.shape
    n = int(n_frames)
    if t >= n:
        start = (t - n) // 2
        return log_mel[:, start:start + n].astype(np.float32)
    out = np.full((n_mels, n), fill_value=MEL_DB_MIN, dtype=np.float32)
    out[:, :t] = log_mel
    return out


def descriptor32_from_logmel(log_mel: np.ndarray) -> np.ndarray:
    """reference :142-149: mean and standard deviation over time of 16 evenly spaced mel bands, standardised."""
    band = log_mel[np.linspace(0, log_mel.shape[0] - 1, 16, dtype=np.int64), :]
    d = np.concatenate([band.mean(axis=1), band.std(axis=1)], axis=0).astype(np.float32)
    return (d - d.mean()) / (d.std() + 1e-8)


@dataclass
class EncoderConfig:
    """reference :152-156."""
    sample_rate: int
    chunk_seconds: float
    z_dim: int


class FrozenLab1Encoder:
    """reference lab3_bridge.py:159-234: the Lab1 checkpoint's encoder, frozen, on the GPU.  The checkpoint ({"cfg", "source_to_idx",
    "model"}: plain containers and tensors) is read with weights_only=True; a GPU is required (no CPU path).  `infer_file_with_mel`
    (audio decoding through librosa) is not provided."""

    def __init__(self, checkpoint_path: Path, device: str = "auto", precision: str = "fp32"):
        self.checkpoint_path = Path(checkpoint_path)
        if not self.checkpoint_path.exists():
            raise FileNotFoundError(f"Checkpoint not found: {self.checkpoint_path}")
        self.device = "cuda" if device == "auto" else device
        payload = torch.load(str(self.checkpoint_path), map_location="cpu", weights_only=True)
        cfg = payload.get("cfg", {})
        source_to_idx = payload.get("source_to_idx", {})
        if not source_to_idx:
            raise ValueError("Checkpoint missing source_to_idx")
        self.cfg = EncoderConfig(sample_rate=int(cfg.get("sample_rate", 22050)), chunk_seconds=float(cfg.get("chunk_seconds", 5.0)),
                                 z_dim=int(cfg.get("z_dim", 128)))
        self.source_to_idx = source_to_idx
        self.model = ChunkEncoder(n_sources=len(source_to_idx), z_dim=self.cfg.z_dim, precision=precision).to(self.device)
        self.model.load_state_dict(payload["model"], strict=False)
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def infer_log_mel(self, log_mel: np.ndarray) -> Dict[str, np.ndarray]:
        """reference :192-201: one [96, T] dB mel -> {"z_content", "z_style"} as float32 arrays."""
        out = self.model(torch.from_numpy(log_mel).unsqueeze(0).to(self.device, non_blocking=True), grl_lambda=0.0)
        return {k: out[k][0].detach().cpu().numpy().astype(np.float32) for k in ("z_content", "z_style")}

    def infer_file_with_mel(self, path, n_frames: int, start_sec: float = 0.0):
        raise NotImplementedError("audio decoding (librosa) stays with the reference's lab3_bridge.py; pass log-mels to infer_log_mel")

    def forward_log_mel_tensor(self, log_mel: torch.Tensor) -> Dict[str, torch.Tensor]:
        """reference :229-234: differentiable pass, log_mel [B, 96, T] in dB."""
        return self.model(log_mel, grl_lambda=0.0)


__all__ = ["ChunkEncoder", "FrozenLab1Encoder", "EncoderConfig", "grad_reverse", "normalize_log_mel", "denormalize_log_mel",
           "fix_log_mel_frames", "descriptor32_from_logmel"]
