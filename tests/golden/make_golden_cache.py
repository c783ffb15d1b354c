#!/usr/bin/env python3
"""A codec cache WRITTEN BY THE REFERENCE ITSELF, as a fixture for the wire-format parity of gan_amd/lab3_codec_data.py.

Run in the build container only (needs /root/reference and transformers):

    python tests/golden/make_golden_cache.py       # writes tests/golden/ref_cache/ and tests/golden/ref_cache_manifest.json

It imports the reference's `lab 3/src/lab3_codec_data.py` (with empty stub modules for the absent `librosa` / `torchaudio`,
inserted after `transformers` has resolved, as make_golden.py does) and calls its own `save_codec_cache`
(lab3_codec_data.py:196-219) on a small procedurally filled cache (5 items, T = 16 frames, with the optional `mert_feat`),
then reads it back with the reference's `load_codec_cache` / `CachedCodecDataset` (:222-289) and records what those return:
keys, dtypes, shapes and fingerprints.  Only data is committed (npz / csv / json), never reference source.
"""
from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/lab 3")

from transformers import AutoFeatureExtractor, AutoModel, EncodecModel  # noqa: E402,F401

for _name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional", "librosa"):
    if _name not in sys.modules:
        sys.modules[_name] = types.ModuleType(_name)

from src import lab3_codec_data as RD  # noqa: E402

from oracle.procfill import proc_input  # noqa: E402

N, T = 5, 16


def cache_arrays():
    """The procedural content of the fixture (tests regenerate it from the same calls)."""
    codes = (proc_input((N, 8, T), "cache.codes", 0, 1.0).numpy() * 300.0 + 512.0).clip(0, 1023).astype(np.int16)
    return {"q_emb": proc_input((N, 128, T), "cache.q_emb", 0, 0.9).numpy(), "codes": codes,
            "z_content": proc_input((N, 128), "cache.zc", 0).numpy(), "z_style": proc_input((N, 128), "cache.zs", 0).numpy(),
            "genre_idx": np.array([0, 1, 2, 1, 0], dtype=np.int64), "mert_feat": proc_input((N, 768), "cache.mert", 0).numpy()}


def main():
    out = HERE / "ref_cache"
    arrays = cache_arrays()
    index_df = pd.DataFrame({"path": [f"clip_{i}.wav" for i in range(N)], "genre": ["rock", "jazz", "hiphop", "jazz", "rock"],
                             "start_sec": [0.0, 5.0, 2.5, 0.0, 7.5]})
    meta = RD.CodecCacheMeta(codec_model_id="facebook/encodec_24khz", codec_sample_rate=24000, codec_chunk_seconds=5.0,
                             codec_bandwidth=6.0, codec_frames=T, codec_channels=128, lab1_n_frames=256)
    RD.save_codec_cache(out, index_df, arrays, {"rock": 0, "jazz": 1, "hiphop": 2}, meta)
    # what the reference itself reads back
    idx2, arr2, gmap2, meta2 = RD.load_codec_cache(out)
    ds = RD.CachedCodecDataset(arr2, np.array([3, 0, 4]))
    item = ds[1]
    with np.load(out / "codec_cache_arrays.npz") as z:
        on_disk = {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape)} for k in z.files}
    manifest = {
        "files": sorted(p.name for p in out.iterdir()),
        "npz_on_disk": on_disk,
        "loaded": {k: {"dtype": str(v.dtype), "shape": list(v.shape), "sum": float(np.asarray(v, dtype=np.float64).sum())}
                   for k, v in arr2.items()},
        "genre_to_idx": gmap2, "meta": meta2.__dict__, "index_columns": list(idx2.columns), "index_rows": int(len(idx2)),
        "dataset_len": len(ds),
        "dataset_item1": {k: {"dtype": str(v.dtype), "shape": list(v.shape), "sum": float(v.double().sum())} for k, v in item.items()},
        "exemplar_index": {str(k): v.tolist() for k, v in RD.build_genre_exemplar_index(arr2["genre_idx"]).items()},
        "generator": "tests/golden/make_golden_cache.py",
    }
    (HERE / "ref_cache_manifest.json").write_text(json.dumps(manifest, indent=1))
    print("wrote", out, sorted(p.name for p in out.iterdir()), sum(p.stat().st_size for p in out.iterdir()), "bytes")


if __name__ == "__main__":
    main()
