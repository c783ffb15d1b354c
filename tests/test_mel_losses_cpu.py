"""CPU: the mel track's stage-2 penalties (gan_amd/lab3_losses.py; plain tensor expressions, no kernel involved) and the host helpers of
gan_amd/lab3_train.py against vectors recorded from the reference's own functions (tests/golden/melstage2_golden.npz, written by
tests/golden/make_golden_melstage2.py from lab 3/src/lab3_losses.py:10-13,74-286 and lab3_train.py:129-204,285-289)."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).parent))
import mel_stage2_common as C  # noqa: E402
from gan_amd import lab3_losses as LL  # noqa: E402
from gan_amd import lab3_train as PT  # noqa: E402
from oracle.procfill import proc_input  # noqa: E402

GOLD = np.load(Path(__file__).parent / "golden" / "melstage2_golden.npz")


def fp(t):
    d = t.detach().double().reshape(-1)
    step = max(1, d.numel() // 8)
    return np.concatenate([[d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], d[::step][:8].numpy()])


def _inputs():
    db = proc_input((3, 32, 64), "melstage2.pen.db", 0, 40.0) - 40.0
    db[:, 20:, :] += 15.0
    db = db.clamp(-80.0, 0.0)
    db2 = (proc_input((3, 32, 64), "melstage2.pen.db2", 0, 40.0) - 40.0).clamp(-80.0, 0.0)
    return db, db2, proc_input((3, 32, 64), "melstage2.pen.a", 0), proc_input((3, 32, 64), "melstage2.pen.b", 0)


def test_db_penalties_match_the_reference_in_value_and_gradient():
    db, db2, a, b = _inputs()
    np.testing.assert_allclose(fp(db), GOLD["pen/in_db_fp"], rtol=1e-12)
    cases = {
        "flatness_closed": (db, lambda x: LL.spectral_flatness_penalty_from_db(x)),
        "flatness": (proc_input((3, 32, 64), "melstage2.pen.flat", 0, 6.0) - 30.0, lambda x: LL.spectral_flatness_penalty_from_db(x)),
        "timbre": (db, lambda x: LL.timbre_balance_penalty_from_db(x, hf_max_ratio=0.2, lf_min_ratio=0.3)),
        "tilt": (db, lambda x: LL.spectral_tilt_penalty_from_db(x, hf_to_lf_max_ratio=0.3)),
        "zcr": (db, lambda x: LL.zcr_proxy_penalty_from_db(x, target_max=0.01)),
        "hf_excess": (db, lambda x: LL.hf_excess_penalty_from_db(x, db2, top_frac=0.25, max_ratio_multiplier=0.5)),
        "freq_l1": (a, lambda x: LL.frequency_weighted_l1_loss(x, b, split_bin=20, low_gain=5.0, high_gain=0.5)),
        "highpass_l1": (a, lambda x: LL.highpass_recon_l1_loss(x, b, low_cut_bin=6, gain=2.0)),
        "cosine": (a, lambda x: LL.cosine_distance(x[:, 0, :], b[:, 0, :]).mean()),
    }
    active = 0
    for name, (src, fn) in cases.items():
        x = src.clone().requires_grad_(True)
        v = fn(x)
        np.testing.assert_allclose(float(v.detach()), float(GOLD[f"pen/{name}/value"]), rtol=2e-6, atol=1e-9, err_msg=name)
        g, = torch.autograd.grad(v, x, allow_unused=True)
        want = GOLD[f"pen/{name}/grad_fp"]
        got = fp(g if g is not None else torch.zeros_like(x))
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-9 + 1e-6 * abs(want[1]), err_msg=name + " gradient")
        active += int(float(v.detach()) > 0)
    assert active == len(cases) - 1, "the case is meant to have the hinges of the penalties open"


def test_band_masks_and_keep_bins():
    db = _inputs()[0]
    np.testing.assert_allclose(fp(LL.lowpass_mel_db(db, keep_bins=24)), GOLD["pen/lowpass_fp"], rtol=1e-7)
    np.testing.assert_allclose(fp(LL.lowpass_mel_db(db, keep_bins=400)), GOLD["pen/lowpass_all_fp"], rtol=1e-7)
    np.testing.assert_allclose(fp(LL.bandpass_mel_db(db, low_bin=4, high_bin=20)), GOLD["pen/bandpass_fp"], rtol=1e-7)
    np.testing.assert_allclose(fp(LL.bandpass_mel_db(db, low_bin=0, high_bin=99)), GOLD["pen/bandpass_edge_fp"], rtol=1e-7)
    assert [LL.estimate_lowpass_keep_bins(n, sr, c) for n, sr, c in C.KEEP_BIN_ARGS] == GOLD["pen/keep_bins"].tolist()
    x = db.clone().requires_grad_(True)
    LL.lowpass_mel_db(x, keep_bins=24).sum().backward()
    assert float(x.grad[:, :24].min()) == 1.0 and float(x.grad[:, 24:].abs().max()) == 0.0  # no gradient into the filled bins
    for fn in (LL.lowpass_mel_db, LL.bandpass_mel_db):
        with pytest.raises(ValueError):
            fn(db[0])
    with pytest.raises(ValueError):
        LL.frequency_weighted_l1_loss(db, db[:, :5])
    with pytest.raises(ValueError):
        LL.hf_excess_penalty_from_db(db, db[:, :5])


def test_bank_builders_and_target_samplers(tmp_path):
    p = tmp_path / "centroids.json"
    p.write_text(json.dumps({"rock": {"vector": [3.0, 4.0]}, "jazz": {"vector160": [0.0, 2.0]}}))
    cm = PT.load_target_centroids(p)
    np.testing.assert_allclose(cm["rock"], [0.6, 0.8], rtol=1e-6)
    np.testing.assert_allclose(cm["jazz"], [0.0, 1.0], rtol=1e-6)
    bank = PT.build_condition_bank({"jazz": 0, "rock": 1}, cm)
    np.testing.assert_allclose(bank.numpy(), [[0.0, 1.0], [0.6, 0.8]], rtol=1e-6)
    with pytest.raises(ValueError):
        PT.build_condition_bank({"pop": 0}, cm)
    with pytest.raises(FileNotFoundError):
        PT.load_target_centroids(tmp_path / "absent.json")
    p.write_text(json.dumps({"rock": {"other": [1.0]}}))
    with pytest.raises(ValueError):
        PT.load_target_centroids(p)
    z = np.asarray([[1.0, 0.0], [0.0, 1.0], [2.0, 0.0]], dtype=np.float32)
    g = np.asarray([0, 0, 2])
    cb = PT.build_style_centroid_bank(z, g, 3).numpy()
    np.testing.assert_allclose(cb, [[0.5 / np.hypot(0.5, 0.5)] * 2, [0.0, 0.0], [1.0, 0.0]], rtol=1e-6)
    eb = PT.build_style_exemplar_bank(z, g, 3)
    assert sorted(eb) == [0, 2] and tuple(eb[0].shape) == (2, 2)
    np.testing.assert_allclose(eb[2].numpy(), [[1.0, 0.0]], rtol=1e-6)
    src = torch.tensor([0, 1, 2, 0, 1, 2])
    assert PT._sample_shift_targets_balanced(src, 3, offset=0).tolist() == [1, 2, 0, 1, 2, 0]
    assert PT._sample_shift_targets_balanced(src, 3, offset=1).tolist() == [1, 2, 0, 1, 2, 0]
    assert PT._sample_shift_targets_balanced(src, 3, offset=2).tolist() == [2, 0, 1, 2, 0, 1]
    t = PT._sample_shift_targets(src.repeat(50), 3)
    assert bool((t != src.repeat(50)).all()) and set(t.tolist()) == {0, 1, 2}
    assert PT._linear_weight(1.0, 3.0, 1, 3) == 2.0 and PT._linear_weight(1.0, 3.0, 0, 1) == 3.0


def test_stage_config_has_the_reference_fields():
    # the history row and the configuration fields the golden run was made with exist under the same names
    kw, _, _ = C.CASES["all_terms"]
    cfg = PT.StageTrainConfig(weights=PT.TrainWeights(**C.ALL_WEIGHTS), **kw)
    assert cfg.cond_mode == "mix" and cfg.style_thaw_scope == "style_head" and cfg.batch_infonce_temp == 0.15
    hist_keys = {k.split("/")[-1] for k in GOLD.files if k.startswith("stage2/all_terms/hist/1/")}
    assert hist_keys == set(PT.STAGE2_KEYS) | {"epoch", "d_steps_taken", "d_steps_skipped", "d_step_rate"}
