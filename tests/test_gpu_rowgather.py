"""GPU: ga_rowgather_fwd / ga_rowgather_bwd (the kernel-row gather that turns a channels-last 2-D convolution into a 1-D implicit GEMM,
gan_amd/conv2d.py) against the tensor-op form it replaced (zero pad, strided row slices, concatenation) and its autograd adjoint, and
conv2d_nhwc / conv_transpose2d_k4s2p1_nhwc against torch's own 2-D convolutions in value and in every gradient."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_amd import _lib as L
from gan_amd.conv2d import RowGather, conv2d_nhwc, conv_transpose2d_k4s2p1_nhwc

DEV = torch.device("cuda")


def gather_reference(x, kh, s, p, OH, kc8):
    B, H, W, C = x.shape
    xp = F.pad(x, (0, 0, 0, 0, p, p))
    rows = [xp[:, i:i + s * (OH - 1) + 1:s] for i in range(kh)]
    if kc8 != kh * C:
        rows.append(x.new_zeros(B, OH, W, kc8 - kh * C))
    return torch.cat(rows, dim=-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 12, 9, 16, 4, 2, 1), (3, 7, 5, 1, 4, 2, 1), (2, 6, 4, 8, 3, 1, 1), (1, 10, 3, 12, 7, 1, 3), (2, 9, 6, 32, 1, 1, 0),
                                   (2, 8, 5, 3, 3, 1, 1)],
                         ids=lambda s: "B%d_H%d_W%d_C%d_k%d_s%d_p%d" % s)
def test_row_gather_and_its_adjoint(shape, dtype):
    B, H, W, C, kh, s, p = shape
    OH = (H + 2 * p - kh) // s + 1
    kc8 = (kh * C + 7) // 8 * 8
    torch.manual_seed(0)
    x = torch.randn(B, H, W, C, device=DEV).to(dtype).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = RowGather.apply(x, kh, s, p, OH, kc8)
    want = gather_reference(xr, kh, s, p, OH, kc8)
    assert torch.equal(y, want)  # a copy: bit-identical
    gy = torch.randn_like(y)
    y.backward(gy)
    want.backward(gy)
    tol = 0.0 if dtype == torch.float32 and kh <= s else (1e-6 if dtype == torch.float32 else 2e-2)  # sums of up to kh/s terms, rounded once here
    assert float((x.grad.float() - xr.grad.float()).abs().max()) <= tol * max(1.0, float(gy.abs().max())) * kh
    # twice differentiable (the R1 path): gradient of <gx, v> with respect to gy is the gather of v
    gy2 = torch.randn_like(y).requires_grad_(True)
    x2 = torch.randn(B, H, W, C, device=DEV).to(dtype).requires_grad_(True)
    gx, = torch.autograd.grad(RowGather.apply(x2, kh, s, p, OH, kc8), x2, gy2, create_graph=True)
    v = torch.randn_like(gx)
    (gx.float() * v.float()).sum().backward()
    assert torch.allclose(gy2.grad.float(), gather_reference(v, kh, s, p, OH, kc8).float(), atol=1e-6)


def test_row_gather_rejects_windows_outside_the_image():  # 1 = GA_ERR_ARG
    x = torch.zeros(1, 4, 4, 8, device=DEV)
    y = torch.zeros(1, 4, 4, 24, device=DEV)
    assert L.lib().ga_rowgather_fwd(x.data_ptr(), y.data_ptr(), L.DT_F32, 1, 4, 4, 8, 3, 1, 0, 4, 24, L.stream()) == 1  # OH too large for pad 0
    assert L.lib().ga_rowgather_fwd(x.data_ptr(), y.data_ptr(), L.DT_F32, 1, 4, 4, 8, 3, 1, 1, 4, 16, L.stream()) == 1  # KC < KH * C
    assert L.lib().ga_rowgather_fwd(0, y.data_ptr(), L.DT_F32, 1, 4, 4, 8, 3, 1, 1, 4, 24, L.stream()) == 1


@pytest.mark.parametrize("case", [(16, 24, 4, 2, 1), (1, 16, 4, 2, 1), (8, 8, 3, 1, 1), (12, 1, 3, 1, 1), (8, 16, 7, 1, 3), (8, 8, 1, 1, 0)],
                         ids=lambda c: "C%d_M%d_k%d_s%d_p%d" % c)
def test_conv2d_matches_torch_in_value_and_gradients(case):
    C, M, k, s, p = case
    torch.manual_seed(1)
    x = torch.randn(2, C, 12, 20, device=DEV)
    w = (torch.randn(M, C, k, k, device=DEV) / (C * k * k) ** 0.5).requires_grad_(True)
    b = (0.1 * torch.randn(M, device=DEV)).requires_grad_(True)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    y = conv2d_nhwc(xa.permute(0, 2, 3, 1).contiguous(), w, b, s, p).permute(0, 3, 1, 2)
    want = F.conv2d(xb, w, b, stride=s, padding=p)
    assert torch.allclose(y, want, rtol=1e-4, atol=1e-5)
    gy = torch.randn_like(want)
    got = torch.autograd.grad(y, [xa, w, b], gy)
    ref = torch.autograd.grad(want, [xb, w, b], gy)
    for g, r, name in zip(got, ref, ("x", "weight", "bias")):
        assert torch.allclose(g, r, rtol=2e-4, atol=2e-5), (name, float((g - r).abs().max()))


def test_conv_transpose2d_matches_torch_in_value_and_gradients():
    torch.manual_seed(2)
    x = torch.randn(2, 16, 6, 10, device=DEV)
    w = (torch.randn(16, 8, 4, 4, device=DEV) / 16.0).requires_grad_(True)
    b = (0.1 * torch.randn(8, device=DEV)).requires_grad_(True)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = conv_transpose2d_k4s2p1_nhwc(xa.permute(0, 2, 3, 1).contiguous(), w, b).permute(0, 3, 1, 2)
    want = F.conv_transpose2d(xb, w, b, stride=2, padding=1)
    assert torch.allclose(y, want, rtol=1e-4, atol=1e-5)
    gy = torch.randn_like(want)
    for g, r, name in zip(torch.autograd.grad(y, [xa, w, b], gy), torch.autograd.grad(want, [xb, w, b], gy), ("x", "weight", "bias")):
        assert torch.allclose(g, r, rtol=2e-4, atol=2e-5), (name, float((g - r).abs().max()))
