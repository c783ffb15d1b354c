"""CPU: every Conv1d / causal reflect conv / ConvTranspose1d forward and data-gradient of the path, expressed as the
generalised forward-form problem that gan_amd.ops hands to ga_conv, equals the stock torch op (via the numpy model
of the C-ABI semantics in tests/emu.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gan_amd import _lib as L
from gan_amd import ops
from gan_amd.ops import ConvGeom
from tests import emu

torch.manual_seed(0)


def _rand(*s):
    return torch.randn(*s, dtype=torch.float64)


CONV_CASES = [
    ConvGeom("conv", 3, 5, 3, 1, 1, 1), ConvGeom("conv", 4, 6, 3, 1, 8, 8), ConvGeom("conv", 1, 4, 15, 1, 7, 1),
    ConvGeom("conv", 3, 4, 41, 4, 20, 1), ConvGeom("conv", 5, 1, 3, 1, 1, 1), ConvGeom("conv", 2, 3, 7, 3, 2, 1),
]


@pytest.mark.parametrize("g", CONV_CASES)
@pytest.mark.parametrize("lin", [37, 64, 75])
def test_conv_forward_and_dgrad(g, lin):
    B = 2
    x = _rand(B, g.cin, lin).requires_grad_(True)
    w = _rand(g.cout, g.cin, g.k)
    b = _rand(g.cout)
    y = F.conv1d(x, w, b, stride=g.stride, padding=g.padding, dilation=g.dilation)
    assert y.shape[-1] == g.out_len(lin)
    p = ops.fwd_params(g, B, lin)
    wp = emu.pack_weights(w.numpy(), None, g.fwd_pack, g.stride, g.padding)
    got = emu.conv(p, x.detach().permute(0, 2, 1).numpy(), wp, bias=b.numpy())
    np.testing.assert_allclose(got, y.detach().permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)
    gy = _rand(*y.shape)
    gx, = torch.autograd.grad(y, x, gy)
    pd = ops.dgrad_params(g, B, lin)
    wd = emu.pack_weights(w.numpy(), None, g.dgrad_pack, g.stride, g.padding)
    got = emu.conv(pd, gy.permute(0, 2, 1).numpy(), wd)
    np.testing.assert_allclose(got, gx.permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)


def _hf_causal_conv(x, w, b, d):
    pad = (w.shape[-1] - 1) * d
    xp = F.pad(x, (pad, 0), mode="reflect") if pad > 0 else x
    return F.conv1d(xp, w, b, dilation=d)


@pytest.mark.parametrize("k,d", [(7, 1), (3, 1), (1, 1), (3, 2)])
def test_causal_reflect_conv_forward_and_dgrad(k, d):
    B, lin, cin, cout = 2, 23, 3, 4
    g = ConvGeom("cconv", cin, cout, k, 1, 0, d)
    x = _rand(B, cin, lin).requires_grad_(True)
    w, b = _rand(cout, cin, k), _rand(cout)
    y = _hf_causal_conv(F.elu(x), w, b, d)
    p = ops.fwd_params(g, B, lin)
    p.pre_act = L.ACT_ELU
    wp = emu.pack_weights(w.numpy(), None, g.fwd_pack, 1, 0)
    got = emu.conv(p, x.detach().permute(0, 2, 1).numpy(), wp, bias=b.numpy())
    np.testing.assert_allclose(got, y.detach().permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)
    # data gradient incl. ELU' and the reflect-pad fold (main launch + in_stride=-1 fix-up, as ops.conv_dgrad does)
    gy = _rand(*y.shape)
    gx, = torch.autograd.grad(y, x, gy)
    xn = x.detach().permute(0, 2, 1).numpy()
    pd = ops.dgrad_params(g, B, lin)
    pd.dact_mode = L.DACT_ELU_FROM_IN
    wd = emu.pack_weights(w.numpy(), None, g.dgrad_pack, 1, 0)
    gyn = gy.permute(0, 2, 1).numpy()
    dx = emu.conv(pd, gyn, wd, dact_src=xn)
    P = min((k - 1) * d, lin - 1)
    if P > 0:
        q = ops._new_params()
        ops._fill(q, B=B, Lin=lin, Cin=cout, M=cin, Lout=P, LoutTot=lin, J=k, in_stride=-1, tap_step=d, off0=0, t0=1,
                  accumulate=1, dact_mode=L.DACT_ELU_FROM_IN)
        dx = emu.conv(q, gyn, wd, y=dx, dact_src=xn)
    np.testing.assert_allclose(dx, gx.permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("r", [8, 5, 4, 2])
def test_causal_convtranspose_forward_and_dgrad(r):
    B, lin, cin, cout = 2, 11, 4, 3
    g = ConvGeom("convt", cin, cout, 2 * r, r)
    x = _rand(B, cin, lin).requires_grad_(True)
    w, b = _rand(cin, cout, 2 * r), _rand(cout)
    y = F.conv_transpose1d(x, w, b, stride=r)
    y = y[..., : y.shape[-1] - r]  # trim_right_ratio = 1: drop k - stride on the right
    assert y.shape[-1] == g.out_len(lin)
    p = ops.fwd_params(g, B, lin)
    wp = emu.pack_weights(w.numpy(), None, g.fwd_pack, r, 0)
    got = emu.conv(p, x.detach().permute(0, 2, 1).numpy(), wp, bias=b.numpy())
    np.testing.assert_allclose(got, y.detach().permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)
    gy = _rand(*y.shape)
    gx, = torch.autograd.grad(y, x, gy)
    pd = ops.dgrad_params(g, B, lin)
    wd = emu.pack_weights(w.numpy(), None, g.dgrad_pack, r, 0)
    got = emu.conv(pd, gy.permute(0, 2, 1).numpy(), wd)
    np.testing.assert_allclose(got, gx.permute(0, 2, 1).numpy(), rtol=1e-10, atol=1e-10)


def test_weight_norm_scale_folds_into_pack():
    g = ConvGeom("convt", 4, 3, 4, 2)
    v, gg = _rand(4, 3, 4), _rand(4, 1, 1).abs() + 0.5
    scale = (gg.reshape(-1) / v.reshape(4, -1).norm(dim=1)).numpy()
    w_eff = v * (gg / v.reshape(4, -1).norm(dim=1).reshape(4, 1, 1))
    a = emu.pack_weights(v.numpy(), scale, g.fwd_pack, 2, 0)
    b = emu.pack_weights(w_eff.numpy(), None, g.fwd_pack, 2, 0)
    np.testing.assert_allclose(a, b, rtol=1e-12)


def test_stft_as_conv_matches_torch_stft():
    """The DFT-basis conv problem built by ops.StftSums equals torch.stft (center, reflect, periodic Hann)."""
    n_fft, hop, S, B = 64, 16, 200, 2
    x = _rand(B, S)
    w_f, w_d, nb, M, J = ops._dft_basis(n_fft, hop, n_fft, torch.float32, "cpu", 1)
    P = n_fft // 2
    rows = (S + 2 * P + hop - 1) // hop
    frames = 1 + S // hop
    xp = np.zeros((B, rows * hop))
    xp[:, : S + 2 * P] = F.pad(x.unsqueeze(1), (P, P), mode="reflect").squeeze(1).numpy()
    p = ops._new_params()
    ops._fill(p, B=B, Lin=rows, Cin=hop, M=M, Lout=frames, LoutTot=frames, J=J, off0=0)
    spec = emu.conv(p, xp.reshape(B, rows, hop), w_f.double().numpy())
    ref = torch.stft(x, n_fft, hop, n_fft, torch.hann_window(n_fft, dtype=torch.float64), return_complex=True, center=True)
    np.testing.assert_allclose(spec[:, :, :nb], ref.real.permute(0, 2, 1).numpy(), atol=1e-5)
    np.testing.assert_allclose(spec[:, :, nb:2 * nb], ref.imag.permute(0, 2, 1).numpy(), atol=1e-5)
    # data-gradient form: d/dxp of sum(spec * G) equals the stride-1 conv of G with the flipped/transposed basis
    G = np.random.RandomState(0).randn(B, frames, M)
    G[:, :, 2 * nb:] = 0
    pd = ops._new_params()
    ops._fill(pd, B=B, Lin=frames, Cin=M, M=hop, Lout=rows, LoutTot=rows, J=J, off0=-(J - 1))
    dxp = emu.conv(pd, G, w_d.double().numpy())
    basis = w_f.double().numpy().reshape(M, n_fft)
    want = np.zeros((B, rows * hop))
    for f in range(frames):
        want[:, f * hop:f * hop + n_fft] += G[:, f, :] @ basis
    np.testing.assert_allclose(dxp.reshape(B, -1), want, atol=1e-4)
