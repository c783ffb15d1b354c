"""CPU: codec-cache wire format (reference lab3_codec_data.py:196-291) and the rank-sharded loader."""
import json

import numpy as np
import pytest
import torch

from gan_amd.lab3_codec_data import (CachedCodecDataset, CodecCacheMeta, ShardedCodecLoader, build_genre_exemplar_index,
                                     load_codec_cache, save_codec_cache)


def _arrays(n=37, c=8, t=11, with_mert=True, seed=0):
    rng = np.random.default_rng(seed)
    a = {"q_emb": rng.standard_normal((n, c, t)).astype(np.float32), "codes": rng.integers(0, 1024, (n, 4, t)).astype(np.int16),
         "z_content": rng.standard_normal((n, 6)).astype(np.float32), "z_style": rng.standard_normal((n, 6)).astype(np.float32),
         "genre_idx": rng.integers(0, 3, n).astype(np.int64)}
    if with_mert:
        a["mert_feat"] = rng.standard_normal((n, 5)).astype(np.float32)
    return a


def test_cache_round_trip_and_file_format(tmp_path):
    import pandas as pd
    arrays = _arrays()
    index_df = pd.DataFrame({"path": [f"a{i}.wav" for i in range(37)], "genre": ["x"] * 37})
    meta = CodecCacheMeta("facebook/encodec_24khz", 24000, 5.0, 6.0, 375, 128, 256)
    save_codec_cache(tmp_path, index_df, arrays, {"rock": 0, "jazz": 1, "pop": 2}, meta)
    # the reference's file names, keys and dtypes (lab3_codec_data.py:203-219)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["codec_cache_arrays.npz", "codec_cache_index.csv",
                                                          "codec_genre_to_idx.json", "codec_meta.json"]
    with np.load(tmp_path / "codec_cache_arrays.npz", allow_pickle=False) as z:
        assert sorted(z.files) == ["codes", "genre_idx", "mert_feat", "q_emb", "z_content", "z_style"]
        assert z["codes"].dtype == np.int16 and z["q_emb"].dtype == np.float32 and z["genre_idx"].dtype == np.int64
    assert json.loads((tmp_path / "codec_meta.json").read_text())["codec_frames"] == 375
    idx2, arr2, g2, meta2 = load_codec_cache(tmp_path)
    assert list(idx2.columns) == ["path", "genre"] and len(idx2) == 37
    assert g2 == {"rock": 0, "jazz": 1, "pop": 2} and meta2 == meta
    for k, v in arrays.items():
        assert arr2[k].dtype == v.dtype and np.array_equal(arr2[k], v), k
    (tmp_path / "codec_meta.json").unlink()
    with pytest.raises(FileNotFoundError):
        load_codec_cache(tmp_path)
    with pytest.raises(KeyError):
        save_codec_cache(tmp_path, index_df, {k: v for k, v in arrays.items() if k != "codes"}, {}, meta)


def test_dataset_items_and_banks():
    arrays = _arrays(with_mert=False)
    ds = CachedCodecDataset(arrays, np.array([5, 1, 30]))
    assert len(ds) == 3
    it = ds[2]
    assert set(it) == {"q_emb", "codes", "z_content", "z_style", "genre_idx"}
    assert it["codes"].dtype == torch.long and it["genre_idx"].dtype == torch.long and it["genre_idx"].dim() == 0
    assert torch.equal(it["q_emb"], torch.from_numpy(arrays["q_emb"][30]))
    ex = build_genre_exemplar_index(arrays["genre_idx"])
    assert sorted(np.concatenate(list(ex.values())).tolist()) == list(range(37))
    assert all(np.all(arrays["genre_idx"][rows] == c) for c, rows in ex.items())


def test_sharded_loader_covers_each_epoch_once_across_ranks():
    arrays = _arrays(n=103)
    ds = CachedCodecDataset(arrays, np.arange(103))
    world, bs = 4, 5
    loaders = [ShardedCodecLoader(ds, bs, rank=r, world_size=world, seed=7, pin=False) for r in range(world)]
    assert all(len(ld) == (103 // bs) // world for ld in loaders)
    for epoch in range(2):
        seen = []
        for ld in loaders:
            ld.set_epoch(epoch)
            for idx in ld.batch_indices():
                assert len(idx) == bs
                seen += idx.tolist()
        assert len(seen) == len(set(seen)) == len(loaders[0]) * world * bs  # disjoint shards, equal batch counts
    # batches carry the rows they name, in the reference's item format, and survive the reuse of the staging buffers
    ld = ShardedCodecLoader(ds, bs, rank=1, world_size=2, seed=3, pin=False, with_codes=True)
    want = [np.sort(i) for i in ld.batch_indices()]
    got = list(ld)
    assert len(got) == len(want) == len(ld)
    for b, rows in zip(got, want):
        assert np.array_equal(b["q_emb"].numpy(), arrays["q_emb"][rows]) and b["q_emb"].dtype == torch.float32
        assert np.array_equal(b["genre_idx"].numpy(), arrays["genre_idx"][rows]) and b["genre_idx"].dtype == torch.long
        assert np.array_equal(b["codes"].numpy(), arrays["codes"][rows].astype(np.int64))
        assert np.array_equal(b["mert_feat"].numpy(), arrays["mert_feat"][rows])
    assert ld.epoch == 1
    with pytest.raises(ValueError):
        ShardedCodecLoader(ds, bs, rank=2, world_size=2)


# ---- wire-format parity against a cache the REFERENCE wrote (tests/golden/make_golden_cache.py) ---------------------------
GOLD = __import__("pathlib").Path(__file__).resolve().parent / "golden"


def _fixture_arrays():
    """The procedural content make_golden_cache.py handed to the reference's save_codec_cache."""
    from oracle.procfill import proc_input
    N, T = 5, 16
    codes = (proc_input((N, 8, T), "cache.codes", 0, 1.0).numpy() * 300.0 + 512.0).clip(0, 1023).astype(np.int16)
    return {"q_emb": proc_input((N, 128, T), "cache.q_emb", 0, 0.9).numpy(), "codes": codes,
            "z_content": proc_input((N, 128), "cache.zc", 0).numpy(), "z_style": proc_input((N, 128), "cache.zs", 0).numpy(),
            "genre_idx": np.array([0, 1, 2, 1, 0], dtype=np.int64), "mert_feat": proc_input((N, 768), "cache.mert", 0).numpy()}


def test_loads_a_cache_written_by_the_reference():
    """load_codec_cache / CachedCodecDataset / build_genre_exemplar_index on the reference-written fixture give what the
    reference's own loader returned for it (recorded in ref_cache_manifest.json) and the arrays that went in."""
    man = json.loads((GOLD / "ref_cache_manifest.json").read_text())
    index_df, arrays, gmap, meta = load_codec_cache(GOLD / "ref_cache")
    want = _fixture_arrays()
    assert set(arrays) == set(man["loaded"]) == set(want)
    for k, rec in man["loaded"].items():
        assert str(arrays[k].dtype) == rec["dtype"] and list(arrays[k].shape) == rec["shape"], k
        assert float(np.asarray(arrays[k], dtype=np.float64).sum()) == pytest.approx(rec["sum"], rel=1e-12, abs=1e-9), k
        np.testing.assert_array_equal(arrays[k], want[k])
    assert gmap == man["genre_to_idx"]
    assert meta.__dict__ == man["meta"]
    assert list(index_df.columns) == man["index_columns"] and len(index_df) == man["index_rows"]
    ds = CachedCodecDataset(arrays, np.array([3, 0, 4]))
    assert len(ds) == man["dataset_len"]
    item = ds[1]
    assert set(item) == set(man["dataset_item1"])
    for k, rec in man["dataset_item1"].items():
        assert str(item[k].dtype) == rec["dtype"] and list(item[k].shape) == rec["shape"], k
        assert float(item[k].double().sum()) == pytest.approx(rec["sum"], rel=1e-12, abs=1e-9), k
    ex = build_genre_exemplar_index(arrays["genre_idx"])
    assert {str(k): v.tolist() for k, v in ex.items()} == man["exemplar_index"]


def test_writes_the_files_the_reference_writes(tmp_path):
    """The reverse direction: save_codec_cache on the same content produces the reference's files - same names, the same npz
    members with the same dtypes and shapes and values, byte-identical JSON files and index CSV."""
    import pandas as pd
    man = json.loads((GOLD / "ref_cache_manifest.json").read_text())
    ref_index = pd.read_csv(GOLD / "ref_cache" / "codec_cache_index.csv")
    save_codec_cache(tmp_path, ref_index, _fixture_arrays(), man["genre_to_idx"], CodecCacheMeta(**man["meta"]))
    assert sorted(p.name for p in tmp_path.iterdir()) == man["files"]
    with np.load(tmp_path / "codec_cache_arrays.npz") as z, np.load(GOLD / "ref_cache" / "codec_cache_arrays.npz") as r:
        assert sorted(z.files) == sorted(r.files) == sorted(man["npz_on_disk"])
        for k, rec in man["npz_on_disk"].items():
            assert str(z[k].dtype) == rec["dtype"] and list(z[k].shape) == rec["shape"], k
            np.testing.assert_array_equal(z[k], r[k])
    for name in ("codec_genre_to_idx.json", "codec_meta.json", "codec_cache_index.csv"):
        assert (tmp_path / name).read_bytes() == (GOLD / "ref_cache" / name).read_bytes(), name
