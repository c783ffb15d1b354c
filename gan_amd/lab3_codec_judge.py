"""MI355X-native drop-in for the reference's `CodecStyleJudge` (`lab 3/src/lab3_codec_judge.py:83-145`), the frozen
classifier whose cross-entropy / embedding / push terms shape the generator's style (`lab3_codec_train.py:480-494`).

Same constructor, `embed` / `forward` / `source_logits_from_embed`, and the same `state_dict()` keys (the parameter
containers are the same torch modules in the same `nn.Sequential` slots).  The three Conv1d + GroupNorm(8) + SiLU stages run
on the HIP kernels behind gan_amd.ops (implicit-GEMM conv, fused GroupNorm+SiLU); the [B, hidden] pooling, the two small
Linear layers and the loss arithmetic on [B, n_genres] logits are plain torch ops.  GPU only.
The judge's own fit loops (AdamW + cosine LR, source-adversary GRL training, INLP) are outside the accelerated path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .lab3_codec_models import _act_dtype, _conv_train, _to_nlc
from .ops import ConvGeom


class _GradientReversal(torch.autograd.Function):
    """reference :68-80"""

    @staticmethod
    def forward(ctx, x, lambd: float):
        ctx.lambd = float(lambd)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.lambd * g, None


class CodecStyleJudge(nn.Module):
    def __init__(self, in_channels: int, n_genres: int, hidden: int = 256, emb_dim: int = 128, n_sources: int = 0,
                 precision: str = "fp32"):
        super().__init__()
        self.in_channels, self.n_genres = int(in_channels), int(n_genres)
        self.hidden, self.emb_dim = int(hidden), int(emb_dim)
        self.n_sources = int(max(0, n_sources))
        self.precision = precision
        self.net = nn.Sequential(
            nn.Conv1d(self.in_channels, self.hidden, kernel_size=5, padding=2), nn.GroupNorm(8, self.hidden), nn.SiLU(),
            nn.Conv1d(self.hidden, self.hidden, kernel_size=3, padding=1), nn.GroupNorm(8, self.hidden), nn.SiLU(),
            nn.Conv1d(self.hidden, self.hidden, kernel_size=3, padding=1), nn.GroupNorm(8, self.hidden), nn.SiLU(),
        )
        self.pool = nn.AdaptiveAvgPool1d(1)  # parameter-free; kept for attribute parity
        self.proj = nn.Linear(self.hidden, self.emb_dim)
        self.head = nn.Linear(self.emb_dim, self.n_genres)
        self.source_head: Optional[nn.Module]
        if self.n_sources > 1:
            src_hidden = max(64, self.emb_dim)
            self.source_head = nn.Sequential(nn.Linear(self.emb_dim, src_hidden), nn.SiLU(), nn.Linear(src_hidden, self.n_sources))
        else:
            self.source_head = None
        self._geoms = (ConvGeom("conv", self.in_channels, self.hidden, 5, 1, 2, 1), ConvGeom("conv", self.hidden, self.hidden, 3, 1, 1, 1),
                       ConvGeom("conv", self.hidden, self.hidden, 3, 1, 1, 1))

    def embed(self, q_emb: torch.Tensor) -> torch.Tensor:
        """q_emb [B, C, T] -> unit-norm embedding [B, emb_dim] (reference :131-135)."""
        L.require_cuda(q_emb)
        if q_emb.dim() != 3 or q_emb.shape[1] != self.in_channels:
            raise ValueError(f"expected q_emb [B, {self.in_channels}, T], got {tuple(q_emb.shape)}")
        dtype = _act_dtype(self.precision)
        h = _to_nlc(q_emb, dtype)
        for i, geom in enumerate(self._geoms):
            conv, gn = self.net[3 * i], self.net[3 * i + 1]
            h = _conv_train(h, conv, geom, dtype)
            h = ops.GnFilmAct.apply(h, gn.weight, gn.bias, None, None, 8, L.ACT_SILU, gn.eps)
        e = self.proj(h.float().mean(dim=1))
        return F.normalize(e, dim=-1)

    def source_logits_from_embed(self, emb: torch.Tensor, grl_lambda: float = 1.0) -> Optional[torch.Tensor]:
        if self.source_head is None:
            return None
        return self.source_head(_GradientReversal.apply(emb, float(grl_lambda)))

    def forward(self, q_emb: torch.Tensor) -> torch.Tensor:
        return self.head(self.embed(q_emb))


def codec_judge_style_terms(style_judge: CodecStyleJudge, q_hat: torch.Tensor, z_style_tgt: torch.Tensor,
                            tgt_genre_idx: torch.Tensor, src_genre_idx: torch.Tensor, embed_align_weight: float,
                            push_margin: float, stage1: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(loss_style, loss_style_embed, loss_style_push) of the G step in mode `codec_judge_ce` (lab3_codec_train.py:480-494)."""
    emb_hat = style_judge.embed(q_hat)
    logits = style_judge.head(emb_hat)
    loss_style = F.cross_entropy(logits, tgt_genre_idx)
    zero = torch.zeros((), device=q_hat.device)
    loss_style_embed = zero
    if float(embed_align_weight) > 0.0 and int(z_style_tgt.shape[1]) == int(emb_hat.shape[1]):
        z_ref = F.normalize(z_style_tgt, dim=-1)
        loss_style_embed = (1.0 - F.cosine_similarity(emb_hat, z_ref, dim=-1)).mean()
        loss_style = loss_style + float(embed_align_weight) * loss_style_embed
    loss_style_push = zero
    if not stage1:
        p_src = torch.softmax(logits, dim=1).gather(1, src_genre_idx.view(-1, 1)).squeeze(1)
        loss_style_push = F.relu(p_src - float(push_margin)).mean()
    return loss_style, loss_style_embed, loss_style_push


__all__ = ["CodecStyleJudge", "codec_judge_style_terms"]
