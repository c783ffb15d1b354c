"""GPU: tensors past 2^31 elements ("max-fit": one MI355X holds batches whose first discriminator feature map and
last decoder activations no longer fit 32-bit element indices, and from B = 512 not even 32-bit BYTE offsets).  Size-independent
property: the batch is copies of 8 distinct items, so every item has a known answer -- its outputs and input gradients must equal
those of its copy among the first eight bit for bit (same kernels, same reduction order per output element), which fails for any kernel
that wraps an index or leaves a 32-bit fast path on too late.  640 discriminator items (2.46e9 elements in the first feature map, 23 GB)
and a 768-item decode (2.9e9 elements in the last residual unit); tools/large_batch_check.py runs 1024 / 1536."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import MultiScaleWaveDiscriminator

DEV = torch.device("cuda")
S, T = 120000, 375


def test_discriminator_forward_and_backward_past_2_31_elements():
    torch.manual_seed(5)
    n = 640
    D = MultiScaleWaveDiscriminator(3, precision="bf16").to(DEV).train()
    base = 0.3 * torch.randn(8, 1, S, device=DEV)
    wav = base.repeat(n // 8, 1, 1).requires_grad_(True)
    outs = D(wav)
    assert outs[0][1][0].numel() > 2 ** 31
    sum(lg.float().mean() for lg, _ in outs).backward()
    torch.cuda.synchronize()
    for si, (lg, feats) in enumerate(outs):
        for k, t in enumerate([lg] + list(feats)):
            assert torch.equal(t[-8:], t[:8]) and torch.equal(t[n // 2:n // 2 + 8], t[:8]), (si, k, tuple(t.shape))
    g = wav.grad
    assert torch.equal(g[-8:], g[:8]) and torch.equal(g[n // 2:n // 2 + 8], g[:8])
    assert bool(torch.isfinite(g[:8]).all()) and float(g[:8].abs().max()) > 0
    gn = torch.stack([p.grad.float().norm() for p in D.parameters() if p.grad is not None])
    assert bool(torch.isfinite(gn).all()) and float(gn.min()) > 0
    del outs, wav, g, D
    torch.cuda.empty_cache()


def test_frozen_decode_past_2_31_elements():
    torch.manual_seed(6)
    n = 768
    codec = FrozenEncodec(random_init=True, device="cuda", precision="bf16")
    qb = 0.5 * torch.randn(8, 128, T, device=DEV)
    with torch.no_grad():
        x = codec.decode_embeddings(qb.repeat(n // 8, 1, 1))
        x8 = codec.decode_embeddings(qb)
    torch.cuda.synchronize()
    assert tuple(x.shape) == (n, 1, S) and bool(torch.isfinite(x).all())
    assert torch.equal(x[-8:], x[:8]) and torch.equal(x[n // 2:n // 2 + 8], x[:8])
    # against an 8-item decode: the LSTM cuts a large batch into groups whose width sets the reduction order
    assert float((x[:8].float() - x8.float()).abs().max() / x8.float().abs().max()) < 2e-2
    del x
    torch.cuda.empty_cache()
