"""GPU: CodecStyleJudge on the HIP kernels against the vectors the reference's own judge produced
(tests/golden/judge_golden.npz) and against the oracle; then the style terms inside train_codec_stage.
fp32: rtol 1e-4.  bf16 (activations and packed weights in bf16, fp32 accumulation; three convs + three GroupNorms):
embedding / logits within 2e-2 of their scale, the loss terms within 2e-2, the gradient within 6e-2 of its max."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_judge import CodecStyleJudge, codec_judge_style_terms
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.lab3_codec_train import (CodecStageTrainConfig, build_q_exemplar_bank, build_style_centroid_bank,
                                      build_style_exemplar_bank, train_codec_stage)
from oracle import codec_oracle as O
from tests.test_style_judge_cpu import ALIGN_W, MARGIN, jgold, judge_case  # noqa: F401  (jgold is a fixture)


def _product(jgold, tag, precision):
    B, T, n_genres, hidden, emb = (int(v) for v in jgold[f"{tag}/cfg"])
    sd, q, zs, tgt, src = judge_case(jgold, tag)
    j = CodecStyleJudge(128, n_genres, hidden=hidden, emb_dim=emb, precision=precision)
    j.load_state_dict(sd, strict=True)
    return j.cuda().eval(), sd, q, zs, tgt, src


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_style_terms_match_reference(jgold, tag, precision):
    j, sd, q, zs, tgt, src = _product(jgold, tag, precision)
    for p in j.parameters():
        p.requires_grad = False
    qd = q.cuda().requires_grad_(True)
    emb = j.embed(qd)
    logits = j.head(emb)
    ls, le, lp = codec_judge_style_terms(j, qd, zs.cuda(), tgt.cuda(), src.cuda(), ALIGN_W, MARGIN, stage1=False)
    g, = torch.autograd.grad(2.0 * ls + 0.7 * lp, qd)
    tol = 1e-4 if precision == "fp32" else 2e-2
    np.testing.assert_allclose(emb.detach().cpu().numpy(), jgold[f"{tag}/emb"], rtol=tol, atol=tol * float(np.abs(jgold[f"{tag}/emb"]).max()))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), jgold[f"{tag}/logits"], rtol=tol,
                               atol=tol * float(np.abs(jgold[f"{tag}/logits"]).max()))
    for got, key in ((ls, "loss_style"), (le, "loss_style_embed"), (lp, "loss_style_push")):
        want = float(jgold[f"{tag}/{key}"])
        assert abs(float(got) - want) <= tol * max(1.0, abs(want)), (key, float(got), want)
    gw = jgold[f"{tag}/grad_q"]
    gtol = 1e-4 if precision == "fp32" else 6e-2
    err = np.abs(g.cpu().numpy() - gw).max()
    assert err <= gtol * float(np.abs(gw).max()), (err, float(np.abs(gw).max()))
    assert j(qd).shape == logits.shape


def test_style_terms_match_oracle_on_a_ragged_case(jgold):
    """Odd T, batch 5, stage1 rule (no push term), alignment weight off: a case the golden file does not hold."""
    torch.manual_seed(5)
    j, sd, _, _, _, _ = _product(jgold, "small", "fp32")
    q = 0.7 * torch.randn(5, 128, 53)
    zs = torch.randn(5, 128)
    tgt = torch.tensor([0, 4, 2, 2, 1])
    src = torch.tensor([1, 1, 3, 0, 4])
    for stage1, aw in ((True, 0.0), (False, 0.25)):
        qo = q.clone().requires_grad_(True)
        ols, ole, olp, oemb, _ = O.style_judge_terms(sd, qo, zs, tgt, src, aw, 0.1, stage1=stage1)
        og, = torch.autograd.grad(ols + olp, qo)
        qd = q.cuda().requires_grad_(True)
        ls, le, lp = codec_judge_style_terms(j, qd, zs.cuda(), tgt.cuda(), src.cuda(), aw, 0.1, stage1=stage1)
        g, = torch.autograd.grad(ls + lp, qd)
        for a, b in ((ls, ols), (le, ole), (lp, olp)):
            assert abs(float(a) - float(b)) <= 1e-4 * max(1.0, abs(float(b)))
        assert (g.cpu() - og).abs().max() <= 1e-4 * og.abs().max()


def test_train_codec_stage_with_style_judge(tmp_path):
    torch.manual_seed(9)
    dev = torch.device("cuda")
    n_genres, T, n, bs = 3, 16, 6, 2
    rng = np.random.default_rng(2)
    q = (0.5 * rng.standard_normal((n, 128, T))).astype(np.float32)
    zc = rng.standard_normal((n, 128)).astype(np.float32)
    zs = rng.standard_normal((n, 128)).astype(np.float32)
    gi = (np.arange(n) % n_genres).astype(np.int64)
    batches = [{"q_emb": torch.from_numpy(q[i:i + bs]), "z_content": torch.from_numpy(zc[i:i + bs]),
                "z_style": torch.from_numpy(zs[i:i + bs]), "genre_idx": torch.from_numpy(gi[i:i + bs])} for i in range(0, n, bs)]
    G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="bf16").to(dev).train()
    D = MultiScaleWaveDiscriminator(2, precision="bf16").to(dev).train()
    codec = FrozenEncodec(random_init=True, device="cuda", precision="bf16")
    judge = CodecStyleJudge(128, n_genres, hidden=64, emb_dim=128, precision="bf16").to(dev)
    before = {k: v.clone() for k, v in judge.state_dict().items()}
    cfg = CodecStageTrainConfig(stage_name="stage2", epochs=1, style_loss_mode="codec_judge_ce", style_embed_align_weight=0.5,
                                style_push_margin=0.05, weights=CodecTrainWeights(content=0.0, style=1.0, style_push=0.5))
    rows = train_codec_stage(cfg, G, D, codec, lab1_encoder=None, style_judge=judge, train_loader=batches, n_genres=n_genres,
                             style_centroid_bank=build_style_centroid_bank(zs, gi, n_genres),
                             style_exemplar_bank=build_style_exemplar_bank(zs, gi, n_genres),
                             q_exemplar_bank=build_q_exemplar_bank(q, gi, n_genres), out_ckpt_dir=tmp_path, device=dev)
    r = rows[0]
    assert r["loss_style"] > 0.5 and r["loss_style_embed"] > 0.0 and r["loss_style_push"] > 0.0 and np.isfinite(r["loss_g"])
    assert not judge.training and all(not p.requires_grad and p.grad is None for p in judge.parameters())
    assert all(torch.equal(v, before[k]) for k, v in judge.state_dict().items())
    # a mode that needs the Lab1 encoder, with a nonzero style weight and no encoder: refused (it would silently drop the term)
    cfg.style_loss_mode = "lab1_cos"
    with pytest.raises(ValueError, match="lab1_encoder is None"):
        train_codec_stage(cfg, G, D, codec, lab1_encoder=None, style_judge=judge, train_loader=batches, n_genres=n_genres,
                          style_centroid_bank=build_style_centroid_bank(zs, gi, n_genres),
                          style_exemplar_bank=build_style_exemplar_bank(zs, gi, n_genres),
                          q_exemplar_bank=build_q_exemplar_bank(q, gi, n_genres), out_ckpt_dir=tmp_path, device=dev)
