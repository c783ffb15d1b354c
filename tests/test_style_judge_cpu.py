"""CPU: the oracle's style-judge restatement against the vectors the reference's own CodecStyleJudge produced
(tests/golden/make_golden_judge.py), and the host-side shape of the product mirror (state_dict keys, argument checks)."""
import numpy as np
import pytest
import torch

from oracle import codec_oracle as O
from oracle.procfill import fill_state_dict, proc_input
from oracle.shapes import style_judge_shapes
from tests.conftest import REPO

ALIGN_W, MARGIN = 0.5, 0.30  # as in make_golden_judge.py


@pytest.fixture(scope="module")
def jgold():
    with np.load(REPO / "tests" / "golden" / "judge_golden.npz") as z:
        return {k: z[k] for k in z.files}


def judge_case(jgold, tag):
    B, T, n_genres, hidden, emb = (int(v) for v in jgold[f"{tag}/cfg"])
    sd = fill_state_dict({k: torch.zeros(s) for k, s in style_judge_shapes(128, n_genres, hidden, emb).items()}, 11)
    q = proc_input((B, 128, T), f"judge.{tag}.q", 0, 0.9)
    zs = proc_input((B, emb), f"judge.{tag}.zs", 0)
    return sd, q, zs, torch.from_numpy(jgold[f"{tag}/tgt"]), torch.from_numpy(jgold[f"{tag}/src"])


@pytest.mark.parametrize("tag", ["small", "full"])
def test_oracle_style_judge_matches_reference(jgold, tag):
    sd, q, zs, tgt, src = judge_case(jgold, tag)
    assert sorted(sd.keys()) == list(jgold[f"{tag}/state_keys"])
    q = q.requires_grad_(True)
    ls, le, lp, emb, logits = O.style_judge_terms(sd, q, zs, tgt, src, ALIGN_W, MARGIN)
    np.testing.assert_allclose(emb.detach().numpy(), jgold[f"{tag}/emb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logits.detach().numpy(), jgold[f"{tag}/logits"], rtol=1e-5, atol=1e-6)
    for got, key in ((ls, "loss_style"), (le, "loss_style_embed"), (lp, "loss_style_push")):
        assert abs(float(got) - float(jgold[f"{tag}/{key}"])) <= 1e-6 * max(1.0, abs(float(jgold[f"{tag}/{key}"]))), key
    g, = torch.autograd.grad(2.0 * ls + 0.7 * lp, q)
    np.testing.assert_allclose(g.numpy(), jgold[f"{tag}/grad_q"], rtol=1e-4, atol=1e-6 * float(np.abs(jgold[f"{tag}/grad_q"]).max()))


def test_oracle_stage1_has_no_push_term(jgold):
    sd, q, zs, tgt, src = judge_case(jgold, "small")
    ls, le, lp, _, _ = O.style_judge_terms(sd, q, zs, tgt, src, 0.0, MARGIN, stage1=True)
    assert float(lp) == 0.0 and float(le) == 0.0 and float(ls) > 0.0


def test_product_judge_has_the_reference_state_dict(jgold):
    from gan_amd.lab3_codec_judge import CodecStyleJudge
    j = CodecStyleJudge(128, 10, hidden=256, emb_dim=128)
    assert sorted(j.state_dict().keys()) == list(jgold["full/state_keys"])
    assert {k: tuple(v.shape) for k, v in j.state_dict().items()} == style_judge_shapes(128, 10, 256, 128)
    j2 = CodecStyleJudge(128, 10, n_sources=3)
    assert any(k.startswith("source_head.") for k in j2.state_dict())
    assert j.source_logits_from_embed(torch.zeros(2, 128)) is None
    e = torch.randn(2, 128, requires_grad=True)
    out = j2.source_logits_from_embed(e, grl_lambda=0.5)  # plain torch Linear layers: runs on the host
    out.sum().backward()
    ref = j2.source_head(e.detach().requires_grad_(True))
    assert out.shape == (2, 3) and torch.allclose(out, ref)
    with pytest.raises(Exception):  # no CPU path for the conv trunk
        j(torch.zeros(1, 128, 8))
