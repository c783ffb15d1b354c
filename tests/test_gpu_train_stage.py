"""GPU: train_codec_stage end to end on a tiny synthetic cache (reference lab3_codec_train.py:267-648): history rows,
checkpoint format (A14, :225-264) and resume."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.lab3_codec_train import (CodecStageTrainConfig, build_q_exemplar_bank, build_style_centroid_bank,
                                      build_style_exemplar_bank, train_codec_stage)

ROW_KEYS = {"stage", "epoch", "loss_g", "loss_d", "loss_adv", "loss_feature_match", "loss_latent_l1", "loss_continuity",
            "loss_content", "loss_style", "loss_style_embed", "loss_style_push", "loss_delta_budget", "loss_mrstft",
            "loss_mode_seeking", "d_steps_taken", "d_steps_skipped", "d_step_rate", "cond_alpha"}


def _loader(n, T, n_genres, bs, seed):
    rng = np.random.default_rng(seed)
    q = (0.5 * rng.standard_normal((n, 128, T))).astype(np.float32)
    zc = rng.standard_normal((n, 128)).astype(np.float32)
    zs = rng.standard_normal((n, 128)).astype(np.float32)
    gi = (np.arange(n) % n_genres).astype(np.int64)
    batches = [{"q_emb": torch.from_numpy(q[i:i + bs]), "z_content": torch.from_numpy(zc[i:i + bs]),
                "z_style": torch.from_numpy(zs[i:i + bs]), "genre_idx": torch.from_numpy(gi[i:i + bs])} for i in range(0, n, bs)]
    return batches, q, zs, gi


@pytest.mark.parametrize("prefetch", [False, True])
def test_train_codec_stage_runs_checkpoints_and_resumes(tmp_path, prefetch):
    torch.manual_seed(328)
    dev = torch.device("cuda")
    n_genres, T = 3, 16
    batches, q, zs, gi = _loader(6, T, n_genres, 2, seed=1)
    G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="bf16").to(dev).train()
    D = MultiScaleWaveDiscriminator(2, precision="bf16").to(dev).train()
    codec = FrozenEncodec(decoder_state_dict=None, device="cuda", precision="bf16")
    cfg = CodecStageTrainConfig(stage_name="stage2", epochs=2, d_step_period=2, r1_gamma=1.0, r1_interval=2,
                                weights=CodecTrainWeights(content=0.0, style=0.0, mode_seeking=0.1))
    kw = dict(lab1_encoder=None, style_judge=None, train_loader=batches, n_genres=n_genres,
              style_centroid_bank=build_style_centroid_bank(zs, gi, n_genres),
              style_exemplar_bank=build_style_exemplar_bank(zs, gi, n_genres),
              q_exemplar_bank=build_q_exemplar_bank(q, gi, n_genres), out_ckpt_dir=tmp_path, device=dev, prefetch=prefetch)
    rows = train_codec_stage(cfg, G, D, codec, **kw)
    assert len(rows) == 2 and all(set(r) == ROW_KEYS for r in rows)
    assert all(np.isfinite(r["loss_g"]) and np.isfinite(r["loss_d"]) for r in rows)
    assert rows[0]["d_steps_taken"] + rows[0]["d_steps_skipped"] == 3 and 0 < rows[0]["d_step_rate"] < 1
    ckpt = tmp_path / "stage2_latest.pt"
    assert ckpt.exists()
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert {"epoch", "generator", "discriminator", "opt_g", "opt_d"} <= set(sd)
    assert sd["epoch"] == 2
    # the optimizer state has torch.optim.AdamW's layout: a stock AdamW over the same parameters loads it
    ref_opt = torch.optim.AdamW([p for p in G.parameters() if p.requires_grad], lr=cfg.lr_g, betas=cfg.betas)
    ref_opt.load_state_dict(sd["opt_g"])
    # resume: nothing left to do at epochs=2, one more epoch at epochs=3
    assert train_codec_stage(cfg, G, D, codec, resume=True, **kw) == []
    cfg.epochs = 3
    more = train_codec_stage(cfg, G, D, codec, resume=True, **kw)
    assert [r["epoch"] for r in more] == [3]


def test_train_codec_stage_from_a_codec_cache_on_disk(tmp_path):
    """The stage fed from the reference's cache format (gan_amd/lab3_codec_data.py): save -> load -> CachedCodecDataset ->
    ShardedCodecLoader (pinned staging, copies issued by the loader) -> train_codec_stage."""
    import pandas as pd
    from gan_amd.lab3_codec_data import CachedCodecDataset, CodecCacheMeta, ShardedCodecLoader, load_codec_cache, save_codec_cache
    torch.manual_seed(9)
    dev = torch.device("cuda")
    n, n_genres, T, bs = 11, 3, 16, 2
    rng = np.random.default_rng(4)
    arrays = {"q_emb": (0.5 * rng.standard_normal((n, 128, T))).astype(np.float32),
              "codes": rng.integers(0, 1024, (n, 8, T)).astype(np.int16),
              "z_content": rng.standard_normal((n, 128)).astype(np.float32),
              "z_style": rng.standard_normal((n, 128)).astype(np.float32), "genre_idx": (np.arange(n) % n_genres).astype(np.int64)}
    save_codec_cache(tmp_path / "cache", pd.DataFrame({"path": [f"{i}.wav" for i in range(n)]}), arrays,
                     {"a": 0, "b": 1, "c": 2}, CodecCacheMeta("facebook/encodec_24khz", 24000, 5.0, 6.0, T, 128, 256))
    _, arr, gmap, meta = load_codec_cache(tmp_path / "cache")
    assert meta.codec_frames == T and len(gmap) == n_genres
    ds = CachedCodecDataset(arr, np.arange(n))
    loader = ShardedCodecLoader(ds, bs, rank=0, world_size=1, seed=1, device=dev)
    assert len(loader) == n // bs
    G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="bf16").to(dev).train()
    D = MultiScaleWaveDiscriminator(2, precision="bf16").to(dev).train()
    codec = FrozenEncodec(decoder_state_dict=None, device="cuda", precision="bf16")
    cfg = CodecStageTrainConfig(stage_name="stage2", epochs=2, d_step_period=1, r1_gamma=0.0,
                                weights=CodecTrainWeights(content=0.0, style=0.0))
    zs, gi = arr["z_style"], arr["genre_idx"]
    rows = train_codec_stage(cfg, G, D, codec, lab1_encoder=None, style_judge=None, train_loader=loader, n_genres=n_genres,
                             style_centroid_bank=build_style_centroid_bank(zs, gi, n_genres),
                             style_exemplar_bank=build_style_exemplar_bank(zs, gi, n_genres),
                             q_exemplar_bank=build_q_exemplar_bank(arr["q_emb"], gi, n_genres), out_ckpt_dir=tmp_path, device=dev)
    assert len(rows) == 2 and all(set(r) == ROW_KEYS for r in rows)
    assert all(np.isfinite(r["loss_g"]) and np.isfinite(r["loss_d"]) for r in rows)
    assert rows[0]["d_steps_taken"] == len(loader) and loader.epoch == 2
