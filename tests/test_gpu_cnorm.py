"""GPU: the channels-last InstanceNorm2d / BatchNorm2d (+ LeakyReLU 0.2) kernels of the mel track (ga_cnorm_*, gan_amd/conv2d.py
`norm2d_nhwc`) against torch's own modules on the same values: output, input gradient, affine gradients, BatchNorm's running
buffers, eval mode, a mean far from zero (the pivot), channel counts that do not divide the block, and the refusals.
fp32 to 2e-5 of the tensor's scale; bf16 operands against the fp32 result of the same bf16-rounded input to a bf16 step."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_amd import _lib as L
from gan_amd import conv2d as C2

DEV = torch.device("cuda")

SHAPES = [(4, 32, 24, 64), (3, 96, 6, 16), (2, 512, 6, 16), (5, 128, 12, 32), (2, 8, 3, 5), (1, 64, 48, 128)]


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12))


def _make(kind, C, affine=True):
    torch.manual_seed(C)
    m = nn.InstanceNorm2d(C, affine=affine) if kind == "instance" else nn.BatchNorm2d(C, affine=affine)
    if affine:
        with torch.no_grad():
            m.weight.copy_(1.0 + 0.3 * torch.randn(C))
            m.bias.copy_(0.2 * torch.randn(C))
    return m.to(DEV)


def _both(kind, shape, dtype, act, train=True, affine=True, offset=0.0, ref_dtype=torch.float32):
    B, C, H, W = shape
    torch.manual_seed(B * 1000 + C)
    x0 = (torch.randn(B, C, H, W, device=DEV) * (0.5 + torch.rand(1, C, 1, 1, device=DEV)) + offset + torch.randn(1, C, 1, 1, device=DEV)).to(dtype)
    gy = torch.randn(B, C, H, W, device=DEV).to(dtype)
    m_ref, m_hip = _make(kind, C, affine), _make(kind, C, affine)
    if kind == "batch":  # non-trivial running buffers
        for m in (m_ref, m_hip):
            m.running_mean.copy_(0.1 * torch.arange(C, device=DEV) / C)
            m.running_var.copy_(1.0 + 0.5 * torch.arange(C, device=DEV) / C)
    m_ref.train(train), m_hip.train(train)
    # reference: torch's module in fp32 on the same (rounded) input values
    m_ref = m_ref.to(ref_dtype)
    xr = x0.to(ref_dtype).detach().clone().requires_grad_(True)
    yr = m_ref(xr)
    if act:
        yr = F.leaky_relu(yr, 0.2)
    yr.backward(gy.to(ref_dtype))
    # HIP: contiguous channels-last
    xh = x0.permute(0, 2, 3, 1).contiguous().detach().clone().requires_grad_(True)
    assert C2.norm2d_supported(m_hip, x0)
    yh = C2.norm2d_nhwc(m_hip, xh, L.ACT_LRELU if act else L.ACT_NONE)
    yh.backward(gy.permute(0, 2, 3, 1).contiguous())
    torch.cuda.synchronize()
    return (yr, xr.grad, m_ref), (yh.permute(0, 3, 1, 2), xh.grad.permute(0, 3, 1, 2), m_hip)


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
@pytest.mark.parametrize("kind", ["instance", "batch"])
@pytest.mark.parametrize("act", [False, True], ids=["plain", "lrelu"])
def test_fp32_matches_torch(kind, shape, act):
    if shape[1] % 4:
        pytest.skip("fp32 pieces are 4 channels")
    (yr, gxr, mr), (yh, gxh, mh) = _both(kind, shape, torch.float32, act)
    assert _rel(yh, yr) <= 2e-5
    assert _rel(gxh, gxr) <= 5e-5
    assert _rel(mh.weight.grad, mr.weight.grad) <= 5e-5 and _rel(mh.bias.grad, mr.bias.grad) <= 5e-5
    if kind == "batch":
        assert _rel(mh.running_mean, mr.running_mean) <= 1e-5 and _rel(mh.running_var, mr.running_var) <= 1e-5
        assert int(mh.num_batches_tracked) == int(mr.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[1] % 8 == 0], ids=str)
@pytest.mark.parametrize("kind", ["instance", "batch"])
def test_bf16_operands_within_a_bf16_step(kind, shape):
    (yr, gxr, mr), (yh, gxh, mh) = _both(kind, shape, torch.bfloat16, True)
    assert yh.dtype == torch.bfloat16 and gxh.dtype == torch.bfloat16
    assert _rel(yh, yr) <= 6e-3           # one rounding of y (2^-8 relative to the element, elements up to the max)
    assert _rel(gxh, gxr) <= 1.5e-2       # the mask comes from the rounded y; gx is rounded once
    assert _rel(mh.weight.grad, mr.weight.grad) <= 5e-3 and _rel(mh.bias.grad, mr.bias.grad) <= 5e-3
    if kind == "batch":
        assert _rel(mh.running_mean, mr.running_mean) <= 1e-5 and _rel(mh.running_var, mr.running_var) <= 1e-5


def test_mean_far_from_zero_keeps_the_variance():
    """|mean| = 300 std: sum(x^2) - N mean^2 in fp32 loses the variance's third digit (torch's fp32 module on this stack is 0.6-1.6 %
    off here); the pivoted sums do not.  Reference: the same module in float64."""
    for kind in ("instance", "batch"):
        (yr, gxr, _), (yh, gxh, _) = _both(kind, (4, 32, 24, 64), torch.float32, False, offset=300.0, ref_dtype=torch.float64)
        assert _rel(yh, yr) <= 2e-4 and _rel(gxh, gxr) <= 2e-4  # x itself carries 300 * 2^-24 = 2e-5 of a std per element


def test_batchnorm_eval_mode_uses_the_running_statistics():
    (yr, gxr, mr), (yh, gxh, mh) = _both("batch", (3, 64, 12, 32), torch.float32, True, train=False)
    assert _rel(yh, yr) <= 2e-5 and _rel(gxh, gxr) <= 2e-5
    assert _rel(mh.weight.grad, mr.weight.grad) <= 5e-5 and _rel(mh.bias.grad, mr.bias.grad) <= 5e-5
    assert int(mh.num_batches_tracked) == 0 and torch.equal(mh.running_mean, mr.running_mean)


def test_no_affine_parameters():
    (yr, gxr, _), (yh, gxh, _) = _both("instance", (2, 32, 12, 32), torch.float32, True, affine=False)
    assert _rel(yh, yr) <= 2e-5 and _rel(gxh, gxr) <= 5e-5


def test_refusals_fall_back_to_the_torch_module():
    lib = L.lib()
    assert not lib.ga_cnorm_supported(L.DT_BF16, 4, 100, 4, 4)        # 4 bf16 channels: no 16-byte piece
    assert not lib.ga_cnorm_supported(L.DT_F32, 4, 100, 2048, 1)      # 512 pieces per row: more than a block
    assert not lib.ga_cnorm_supported(L.DT_F32, 4, 100, 32, 2)        # groups is B or 1
    x = torch.zeros(4, 100, 4, device=DEV, dtype=torch.bfloat16)
    s = torch.zeros(4, 4, 2, device=DEV, dtype=torch.float64)
    assert lib.ga_cnorm_stats(x.data_ptr(), L.DT_BF16, 4, 100, 4, 4, s.data_ptr(), L.stream()) != 0
    m = nn.BatchNorm2d(32, momentum=None).to(DEV)                     # cumulative average: torch's module keeps it
    assert not C2.norm2d_supported(m, torch.zeros(2, 32, 4, 4, device=DEV))
    with C2.twice_differentiable():
        assert not C2.norm2d_supported(nn.BatchNorm2d(32).to(DEV), torch.zeros(2, 32, 4, 4, device=DEV))


def test_decoder_and_discriminator_agree_with_the_torch_norm_layers(monkeypatch):
    """The mel decoder / discriminator with the fused norm layers against the same modules with torch's norm layers (GA_MEL_FUSED_NORM=0
    path), fp32 mode: output, parameter gradients and BatchNorm buffers."""
    from gan_amd import lab3_models as PM
    torch.manual_seed(3)
    G = PM.ReconstructionDecoder(base_channels=64, precision="fp32").to(DEV).train()
    D = PM.MelDiscriminator(base_channels=16, cond_dim=160, precision="fp32").to(DEV).train()
    G2, D2 = copy.deepcopy(G), copy.deepcopy(D)
    zc, cond = torch.randn(3, 128, device=DEV), torch.randn(3, 160, device=DEV)

    def run(g, d):
        mel = g(zc, cond)
        logit = d(mel, cond)
        (logit.mean() + mel.square().mean()).backward()
        return mel, logit

    mel_a, logit_a = run(G, D)
    monkeypatch.setattr(C2, "FUSED_NORM", False)
    mel_b, logit_b = run(G2, D2)
    assert _rel(mel_a, mel_b) <= 1e-4 and _rel(logit_a, logit_b) <= 1e-4
    for (k, p), (_, q) in zip(list(G.named_parameters()) + list(D.named_parameters()), list(G2.named_parameters()) + list(D2.named_parameters())):
        scale = float(q.grad.abs().max())
        # a conv bias in front of a norm layer has a zero gradient in exact arithmetic: compare against the layer's scale
        assert float((p.grad - q.grad).abs().max()) <= 2e-3 * max(scale, 1e-3), k
    for (k, a), (_, b) in zip(D.named_buffers(), D2.named_buffers()):
        assert _rel(a.float(), b.float()) <= 1e-4, k
