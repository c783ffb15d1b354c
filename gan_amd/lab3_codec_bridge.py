"""MI355X-native frozen EnCodec (SEANet) decoder behind the reference's `FrozenEncodec.decode_embeddings`
(/root/reference/lab 3/src/lab3_codec_bridge.py:135-146).

The reference calls the third-party HF `EncodecDecoder` (transformers modeling_encodec.py:316-347).  Here the
same layer stack runs on the gan_amd convolution kernels in channels-last layout:
  EncodecConv1d (:76-176)          -> ops.ConvFrozen('cconv'): causal reflect pad fused into the window load
  EncodecConvTranspose1d (:179-233)-> ops.ConvFrozen('convt'): one 2-tap GEMM phase per output residue, right trim implicit
  ELU (:268,330,340)               -> fused as the consumer's load prologue (and ELU' in its data-gradient epilogue)
  EncodecResnetBlock (:252-283)    -> shortcut conv, then k3 conv, then k1 conv whose epilogue adds the shortcut
  EncodecLSTM (:236-249)           -> ops.LstmSkip: one k=1 GEMM for the layer-1 input projection + T+1 fused step launches
Weights are frozen; only the gradient w.r.t. the input latents is produced.  `state_dict()` keys equal the HF
decoder's, so `SeanetDecoder.load_state_dict(hf_model.decoder.state_dict())` is the integration step
(loading the checkpoint itself, `EncodecModel.from_pretrained`, stays with the caller: it is a network fetch).
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .lab3_codec_models import _act_dtype
from .ops import ConvGeom


class _WN(nn.Module):
    """Holds parametrizations.weight.original0 / original1 (torch.nn.utils.parametrizations.weight_norm naming)."""

    def __init__(self, g_shape, v_shape):
        super().__init__()
        self.parametrizations = nn.Module()
        self.parametrizations.weight = nn.Module()
        self.parametrizations.weight.original0 = nn.Parameter(torch.ones(g_shape), requires_grad=False)
        self.parametrizations.weight.original1 = nn.Parameter(torch.randn(v_shape) * 0.05, requires_grad=False)

    @property
    def g(self):
        return self.parametrizations.weight.original0

    @property
    def v(self):
        return self.parametrizations.weight.original1


class _ConvLayer(nn.Module):
    def __init__(self, geom: ConvGeom):
        super().__init__()
        d0, d1, k = geom.wdims
        self.conv = _WN((d0, 1, 1), (d0, d1, k))
        self.conv.bias = nn.Parameter(torch.zeros(geom.cout), requires_grad=False)  # key: <layer>.conv.bias
        self.geom = geom


class _ResBlock(nn.Module):
    def __init__(self, dim: int, compress: int, k: int):
        super().__init__()
        hidden = dim // compress
        self.block = nn.Module()
        self.block.add_module("1", _ConvLayer(ConvGeom("cconv", dim, hidden, k)))
        self.block.add_module("3", _ConvLayer(ConvGeom("cconv", hidden, dim, 1)))
        self.shortcut = _ConvLayer(ConvGeom("cconv", dim, dim, 1))


class _LSTM(nn.Module):
    def __init__(self, dim: int, layers: int):
        super().__init__()
        self.lstm = nn.LSTM(dim, dim, layers)
        for p in self.lstm.parameters():
            p.requires_grad = False


class SeanetDecoder(nn.Module):
    """EncodecDecoder with EncodecConfig() defaults (24 kHz): ratios (8,5,4,2), 32 filters, 2 LSTM layers,
    causal, reflect pad, weight_norm, 1 residual layer (dilation 1), compress 2, conv shortcut."""

    def __init__(self, hidden_size: int = 128, num_filters: int = 32, ratios=(8, 5, 4, 2), kernel_size: int = 7,
                 last_kernel_size: int = 7, residual_kernel_size: int = 3, compress: int = 2, lstm_layers: int = 2,
                 precision: str = "fp32"):
        super().__init__()
        self.precision = precision
        self.use_library_lstm = False  # True: torch's LSTM op (MIOpen) instead of the gan_amd step kernels (A/B checks)
        self.fuse_resblock = os.environ.get("GA_FUSE_RESBLOCK", "1") != "0"  # False: a residual unit as three autograd nodes (A/B checks)
        self.fuse_elu = os.environ.get("GA_FUSE_ELU", "1") != "0"            # False: every ELU in front of a wide conv is a pass of its own
        self.fuse_elu2 = os.environ.get("GA_FUSE_ELU2", "1") != "0"          # False: no second (ELU) output from the transposed convs' epilogues
        self.fuse_tail = os.environ.get("GA_FUSE_TAIL", "1") != "0"          # False: no-gradient decodes end with separate unit / last-conv launches
        self.ratios = tuple(int(r) for r in ratios)
        c = (2 ** len(self.ratios)) * num_filters
        layers = nn.Module()
        self._plan: List[Tuple[str, str]] = []
        layers.add_module("0", _ConvLayer(ConvGeom("cconv", hidden_size, c, kernel_size)))
        self._plan.append(("conv", "0"))
        layers.add_module("1", _LSTM(c, lstm_layers))
        self._plan.append(("lstm", "1"))
        idx = 2
        for r in self.ratios:
            idx += 1  # ELU (fused into the transposed conv's load)
            layers.add_module(str(idx), _ConvLayer(ConvGeom("convt", c, c // 2, 2 * r, r)))
            self._plan.append(("elu_convt", str(idx)))
            idx += 1
            c //= 2
            layers.add_module(str(idx), _ResBlock(c, compress, residual_kernel_size))
            self._plan.append(("resblock", str(idx)))
            idx += 1
        idx += 1  # ELU
        layers.add_module(str(idx), _ConvLayer(ConvGeom("cconv", num_filters, 1, last_kernel_size)))
        self._plan.append(("elu_conv", str(idx)))
        self.layers = layers
        self._packs: Dict[Tuple[str, torch.dtype], Tuple[torch.Tensor, torch.Tensor]] = {}
        for p in self.parameters():
            p.requires_grad = False

    # packed weights are cached: the decoder is frozen.  Call after loading / moving the module.
    def invalidate_packs(self) -> None:
        self._packs.clear()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate_packs()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._packs = {}
        return out

    def _pack(self, name: str, layer: _ConvLayer, dtype):
        key = (name, dtype)
        if key not in self._packs:
            self._packs[key] = ops.pack_frozen(layer.geom, layer.conv.v.data, layer.conv.g.data, dtype)
        return self._packs[key]

    def _conv(self, name: str, layer: _ConvLayer, x, dtype, pre_act=L.ACT_NONE, res=None, out_f32=False, also_elu=False):
        wp_f, wp_d = self._pack(name, layer, dtype)
        return ops.ConvFrozen.apply(x, res, wp_f, wp_d, layer.conv.bias.data, layer.geom, pre_act, out_f32, also_elu)

    def forward(self, q_bct: torch.Tensor, n_grad: Optional[int] = None):
        """q [B, 128, T] -> wav [B, 1, T*prod(ratios)].
        With n_grad, the batch is [n_grad items that need d(wav)/d(q) ; frozen items]: the first conv and the
        (launch-bound) LSTM run once over the whole batch, the convolutional tail runs per part, and the result is the
        pair (wav of the gradient part, wav of the frozen part)."""
        L.require_cuda(q_bct)
        dtype = _act_dtype(self.precision)
        h = q_bct.permute(0, 2, 1).to(dtype).contiguous()  # NLC
        if n_grad is not None and not self.use_library_lstm and 0 < n_grad < h.shape[0]:
            h = self._conv("0", getattr(self.layers, "0"), h, dtype)
            key = ("lstm", dtype)
            if key not in self._packs:
                self._packs[key] = ops.prepare_lstm_weights(getattr(self.layers, "1").lstm, dtype)
            h = ops.LstmSkip.apply(h, self._packs[key], int(n_grad))
            out_g = self._tail(h[:n_grad], dtype, self._plan[2:])
            with torch.no_grad():
                out_f = self._tail(h[n_grad:].detach(), dtype, self._plan[2:])
            return out_g, out_f
        if n_grad is not None:
            if n_grad >= h.shape[0]:
                return self._tail(h, dtype, self._plan), None
            out_g = self._tail(h[:n_grad], dtype, self._plan)
            with torch.no_grad():
                out_f = self._tail(h[n_grad:].detach(), dtype, self._plan)
            return out_g, out_f
        return self._tail(h, dtype, self._plan)

    def _tail(self, h: torch.Tensor, dtype, plan) -> torch.Tensor:
        elu_done = False  # h holds ELU(h): written by the residual unit in front (ops.ResBlockFrozen, elu_out)
        h_elu = None      # ELU(h) next to h: written by the transposed conv in front (ga_conv_params.y2) for the unit's first conv
        for pos, (kind, name) in enumerate(plan):
            layer = getattr(self.layers, name)
            if kind == "conv":
                h = self._conv(name, layer, h, dtype)
            elif kind == "lstm":
                if self.use_library_lstm:
                    seq = h.permute(1, 0, 2).float()  # [T, B, C]
                    was_training = layer.lstm.training
                    layer.lstm.train(True)  # lab3_codec_bridge.py:140-145: RNN backward needs train mode
                    out, _ = layer.lstm(seq)
                    layer.lstm.train(was_training)
                    h = (out + seq).permute(1, 0, 2).to(dtype).contiguous()
                else:
                    key = ("lstm", dtype)
                    if key not in self._packs:
                        self._packs[key] = ops.prepare_lstm_weights(layer.lstm, dtype)
                    h = ops.LstmSkip.apply(h, self._packs[key])
            elif kind == "elu_convt":
                nxt = plan[pos + 1] if pos + 1 < len(plan) else None
                two = False
                if dtype == torch.bfloat16 and self.fuse_resblock and self.fuse_elu and self.fuse_elu2 and nxt is not None and nxt[0] == "resblock":
                    # the unit behind reads this layer's output raw (shortcut) and through ELU (first conv): where that ELU would be
                    # a pass of its own (wide units on the gathered-GEMM kernel), this layer's epilogue writes it as well
                    unit = getattr(self.layers, nxt[1])
                    b1g = getattr(unit.block, "1").geom
                    meta = torch.empty((h.shape[0], layer.geom.out_len(h.shape[1]), b1g.cin), dtype=h.dtype, device="meta")
                    two = (not ops.resunit_fusable(meta, unit.shortcut.geom, b1g, getattr(unit.block, "3").geom)
                           and ops.elu_needs_a_pass(b1g, meta) and ops.conv_second_output_ok(layer.geom, h))
                out = self._conv(name, layer, h, dtype, pre_act=ops.ACT_ELU_DONE if elu_done else L.ACT_ELU, also_elu=two)
                h, h_elu = out if two else (out, None)
                elu_done = False
            elif kind == "resblock":
                b1, b3 = getattr(layer.block, "1"), getattr(layer.block, "3")
                nxt = plan[pos + 1] if pos + 1 < len(plan) else None
                if (h.dtype == torch.bfloat16 and self.fuse_resblock and self.fuse_tail and nxt is not None and nxt[0] == "elu_conv"
                        and pos + 2 == len(plan) and not (torch.is_grad_enabled() and h.requires_grad) and h.shape[1] >= 13
                        and ops.resunit_fusable(h, layer.shortcut.geom, b1.geom, b3.geom) and b3.geom.cout == 32):
                    # nobody wants a gradient (the frozen items of a step): the last unit and the closing ELU + conv as one launch,
                    # the unit's output never leaves the chip
                    last = getattr(self.layers, nxt[1])
                    if (last.geom.kind, last.geom.cin, last.geom.cout, last.geom.k, last.geom.dilation) == ("cconv", 32, 1, 7, 1):
                        wav, _ = ops.resunit_forward(
                            h.contiguous(), self._pack(name + ".sc", layer.shortcut, dtype)[0], layer.shortcut.conv.bias.data,
                            self._pack(name + ".b1", b1, dtype)[0], b1.conv.bias.data, self._pack(name + ".b3", b3, dtype)[0],
                            b3.conv.bias.data, want_t=False, last=(self._pack(nxt[1], last, dtype)[0], last.conv.bias.data))
                        return wav.permute(0, 2, 1)
                if h.dtype == torch.bfloat16 and self.fuse_resblock:  # one autograd node: no gradient-sum pass over h
                    # where the conv behind an ELU would need a pass to materialise it (wide layers on the gathered-GEMM kernel),
                    # the producing epilogue writes the ELU instead: inside the unit, and for the transposed conv that follows it
                    elu_mid = self.fuse_elu and ops.elu_needs_a_pass(b3.geom, torch.empty((h.shape[0], h.shape[1], b3.geom.cin), dtype=h.dtype, device="meta"))
                    elu_out = (self.fuse_elu and nxt is not None and nxt[0] == "elu_convt" and b3.geom.cout >= 128
                               and ops.elu_needs_a_pass(getattr(self.layers, nxt[1]).geom, h))
                    h = ops.ResBlockFrozen.apply(
                        h, *self._pack(name + ".sc", layer.shortcut, dtype), layer.shortcut.conv.bias.data, layer.shortcut.geom,
                        *self._pack(name + ".b1", b1, dtype), b1.conv.bias.data, b1.geom,
                        *self._pack(name + ".b3", b3, dtype), b3.conv.bias.data, b3.geom, elu_mid, elu_out, h_elu)
                    elu_done = elu_out
                    h_elu = None
                else:
                    sc = self._conv(name + ".sc", layer.shortcut, h, dtype)
                    t = self._conv(name + ".b1", b1, h, dtype, pre_act=L.ACT_ELU)
                    h = self._conv(name + ".b3", b3, t, dtype, pre_act=L.ACT_ELU, res=sc)
            elif kind == "elu_conv":
                # the waveform leaves the decoder in fp32 even in bf16 mode: the MR-STFT log-magnitude loss reads it
                h = self._conv(name, layer, h, dtype, pre_act=L.ACT_ELU, out_f32=True)
        return h.permute(0, 2, 1)  # [B, 1, S]


@dataclass
class CodecConfig:
    """reference lab3_codec_bridge.py:15-24"""
    model_id: str = "facebook/encodec_24khz"
    sample_rate: int = 24000
    chunk_seconds: float = 5.0
    bandwidth: float = 6.0
    codebook_size: int = 1024
    num_codebooks: int = 8
    latent_channels: int = 128
    frame_rate: float = 75.0


class FrozenEncodec:
    """The part of the reference's FrozenEncodec that sits on the training hot path: `.cfg` and
    `.decode_embeddings(q[B,C,T]) -> wav[B,1,S]`, differentiable w.r.t. q.

    Same constructor arguments as the reference (lab3_codec_bridge.py:33-60: model_id, bandwidth, chunk_seconds, device), so
    `FrozenEncodec(model_id=..., bandwidth=..., chunk_seconds=..., device=...)` keeps working: like the reference it then
    fetches the pretrained model through `transformers.EncodecModel.from_pretrained(model_id)` and copies the decoder's
    state dict into the native decoder.  Extra keywords: `decoder_state_dict` (an HF decoder state dict the caller already
    holds: no transformers call), `random_init=True` (keep the random initialisation: synthetic benchmarks and tests, no
    network), `precision` ("fp32" | "bf16")."""

    def __init__(self, model_id: str = "facebook/encodec_24khz", bandwidth: float = 6.0, chunk_seconds: float = 5.0,
                 device: str = "auto", *, decoder_state_dict: Optional[dict] = None, random_init: bool = False,
                 precision: str = "fp32"):
        if device == "auto":  # reference _to_device (:27-30)
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        self.model_id = str(model_id)
        self.decoder = SeanetDecoder(precision=precision)
        cfg = CodecConfig(model_id=self.model_id, chunk_seconds=float(chunk_seconds), bandwidth=float(bandwidth))
        if decoder_state_dict is None and not random_init:
            from transformers import EncodecModel  # the reference's own loading step (:43): a fetch by model name
            hf = EncodecModel.from_pretrained(self.model_id)
            decoder_state_dict = hf.decoder.state_dict()
            hc = hf.config
            cfg = CodecConfig(model_id=self.model_id, sample_rate=int(getattr(hc, "sampling_rate", 24000)),
                              chunk_seconds=float(chunk_seconds), bandwidth=float(bandwidth), codebook_size=int(hc.codebook_size),
                              num_codebooks=int(hc.num_quantizers), latent_channels=int(getattr(hc, "hidden_size", 128)),
                              frame_rate=float(getattr(hf.quantizer, "frame_rate", None) or 75.0))
        if decoder_state_dict is not None:
            self.decoder.load_state_dict(decoder_state_dict, strict=True)
        self.decoder.to(device)
        self.cfg = cfg

    def decode_embeddings(self, q_emb_b_c_t: torch.Tensor) -> torch.Tensor:
        return self.decoder(q_emb_b_c_t.to(self.device))

    def decode_embeddings_mixed(self, q_grad_b_c_t: torch.Tensor, q_frozen_b_c_t: torch.Tensor):
        """(decode_embeddings(q_grad), decode_embeddings(q_frozen) without gradient) as one pass through the shared,
        launch-bound front of the decoder (first conv + LSTM); not in the reference, which makes four separate calls
        (lab3_codec_train.py:413,415,422,458)."""
        n = int(q_grad_b_c_t.shape[0])
        q = torch.cat([q_grad_b_c_t, q_frozen_b_c_t.detach().to(q_grad_b_c_t.dtype)], dim=0)
        return self.decoder(q.to(self.device), n_grad=n)

    def expected_num_frames(self) -> int:
        return int(round(float(self.cfg.chunk_seconds) * float(self.cfg.frame_rate)))

    def target_num_samples(self) -> int:
        return int(round(float(self.cfg.chunk_seconds) * float(self.cfg.sample_rate)))
