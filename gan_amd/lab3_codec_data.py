"""Codec-cache wire format and loaders (reference `lab 3/src/lab3_codec_data.py:196-291`).

Only the part of the reference module that the training stage consumes is mirrored: the on-disk cache written by
`save_codec_cache` / read by `load_codec_cache` (same four file names, the same array keys and dtypes, the same JSON
fields), `CachedCodecDataset` (same item dictionary), `build_genre_exemplar_index`, and -- new, the MI355X-side piece --
`ShardedCodecLoader`, which feeds `train_codec_stage` from a loaded cache with rank-sharded, pinned, non-blocking
host-to-device batches (one process per GPU: each rank sees its own 1/world slice of every epoch's permutation).
Building a cache from audio (EnCodec encode, Lab1 embeddings, librosa) is outside the accelerated path and is not here.

  codec_cache_arrays.npz   q_emb f32 [N, C, T], codes i16 [N, Q, T], z_content f32 [N, Dc], z_style f32 [N, Ds],
                           genre_idx i64 [N], optional mert_feat f32 [N, Dm]            (np.savez_compressed)
  codec_cache_index.csv    one row per item (free-form columns; kept as a DataFrame)
  codec_genre_to_idx.json  {genre name: index}
  codec_meta.json          CodecCacheMeta fields
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

_ARRAYS = "codec_cache_arrays.npz"
_INDEX = "codec_cache_index.csv"
_GENRES = "codec_genre_to_idx.json"
_META = "codec_meta.json"
_REQUIRED = {"q_emb": np.float32, "codes": np.int16, "z_content": np.float32, "z_style": np.float32, "genre_idx": np.int64}
_OPTIONAL = {"mert_feat": np.float32}


@dataclass
class CodecCacheMeta:
    """reference lab3_codec_data.py:28-36"""
    codec_model_id: str
    codec_sample_rate: int
    codec_chunk_seconds: float
    codec_bandwidth: float
    codec_frames: int
    codec_channels: int
    lab1_n_frames: int


def save_codec_cache(cache_dir: Path, index_df, arrays: Dict[str, np.ndarray], genre_to_idx: Dict[str, int],
                     meta: CodecCacheMeta) -> None:
    """reference :196-219 -- same files, keys and dtypes, so either side reads the other's cache."""
    cache_dir = Path(cache_dir)
    cache_dir.mkdir(parents=True, exist_ok=True)
    missing = [k for k in _REQUIRED if k not in arrays]
    if missing:
        raise KeyError(f"arrays lacks {missing}")
    n = len(arrays["genre_idx"])
    out = {}
    for k, dt in {**_REQUIRED, **_OPTIONAL}.items():
        if k in arrays:
            a = np.asarray(arrays[k])
            if len(a) != n:
                raise ValueError(f"{k} has {len(a)} rows, genre_idx has {n}")
            out[k] = a.astype(dt, copy=False)
    index_df.to_csv(cache_dir / _INDEX, index=False)
    np.savez_compressed(cache_dir / _ARRAYS, **out)
    (cache_dir / _GENRES).write_text(json.dumps({str(k): int(v) for k, v in genre_to_idx.items()}, indent=2), encoding="utf-8")
    (cache_dir / _META).write_text(json.dumps(asdict(meta), indent=2), encoding="utf-8")


def load_codec_cache(cache_dir: Path):
    """reference :222-261 -> (index DataFrame, arrays, genre_to_idx, CodecCacheMeta); FileNotFoundError when a file is missing."""
    import pandas as pd
    cache_dir = Path(cache_dir)
    paths = [cache_dir / f for f in (_INDEX, _ARRAYS, _GENRES, _META)]
    if not all(p.exists() for p in paths):
        raise FileNotFoundError(f"Codec cache files missing in {cache_dir}")
    index_df = pd.read_csv(paths[0])
    with np.load(paths[1], allow_pickle=False) as z:
        arrays = {k: z[k].astype(dt) for k, dt in _REQUIRED.items()}
        for k, dt in _OPTIONAL.items():
            if k in z.files:
                arrays[k] = z[k].astype(dt)
    genre_to_idx = {str(k): int(v) for k, v in json.loads(paths[2].read_text(encoding="utf-8")).items()}
    raw = json.loads(paths[3].read_text(encoding="utf-8"))
    meta = CodecCacheMeta(codec_model_id=str(raw["codec_model_id"]), codec_sample_rate=int(raw["codec_sample_rate"]),
                          codec_chunk_seconds=float(raw["codec_chunk_seconds"]), codec_bandwidth=float(raw["codec_bandwidth"]),
                          codec_frames=int(raw["codec_frames"]), codec_channels=int(raw["codec_channels"]),
                          lab1_n_frames=int(raw["lab1_n_frames"]))
    return index_df, arrays, genre_to_idx, meta


class CachedCodecDataset(Dataset):
    """reference :264-289: a row subset of the cache arrays; items are the same dictionaries."""

    def __init__(self, arrays: Dict[str, np.ndarray], indices: np.ndarray):
        indices = np.asarray(indices, dtype=np.int64)
        self.q_emb = arrays["q_emb"][indices]
        self.codes = arrays["codes"][indices]
        self.z_content = arrays["z_content"][indices]
        self.z_style = arrays["z_style"][indices]
        self.genre_idx = arrays["genre_idx"][indices]
        self.mert_feat = arrays["mert_feat"][indices] if "mert_feat" in arrays else None

    def __len__(self) -> int:
        return int(len(self.genre_idx))

    def __getitem__(self, i: int) -> Dict[str, torch.Tensor]:
        item = {"q_emb": torch.from_numpy(self.q_emb[i]), "codes": torch.from_numpy(self.codes[i]).long(),
                "z_content": torch.from_numpy(self.z_content[i]), "z_style": torch.from_numpy(self.z_style[i]),
                "genre_idx": torch.tensor(int(self.genre_idx[i]), dtype=torch.long)}
        if self.mert_feat is not None:
            item["mert_feat"] = torch.from_numpy(self.mert_feat[i])
        return item


def build_genre_exemplar_index(genre_idx: np.ndarray) -> Dict[int, np.ndarray]:
    """reference :292-297: {class: row indices of that class}."""
    g = np.asarray(genre_idx, dtype=np.int64)
    return {int(c): np.flatnonzero(g == c).astype(np.int64) for c in np.unique(g)}


class ShardedCodecLoader:
    """Batches of a CachedCodecDataset for `train_codec_stage`: the role of DataLoader(shuffle=True, drop_last=True) plus a
    DistributedSampler in the reference's single-process runner, shaped for one process per GPU.

      * every epoch draws ONE permutation from (seed, epoch) -- identical on all ranks -- and rank r takes the batches
        r, r + world, r + 2*world, ... of it: disjoint shards, the same number of batches on every rank (trailing batches
        that would leave some rank without one are dropped, as drop_last does for the ragged tail);
      * a batch is gathered with one fancy-index per array straight into PINNED staging buffers (two sets, used
        alternately) and, when `device` is given, copied to the GPU right there with non-blocking copies on the current
        stream; an event per staging set marks the end of its copies and is waited for before the set is gathered into
        again, so the host never overwrites bytes a queued copy has yet to read (192 KB / sample of q_emb: PCIe is never
        the limit, the copy overlaps the running iteration);
      * items are the reference's dictionaries, batched: q_emb [B, C, T] f32, z_content, z_style [B, D] f32,
        genre_idx [B] i64 (+ codes / mert_feat when asked for)."""

    def __init__(self, dataset: CachedCodecDataset, batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0,
                 shuffle: bool = True, with_codes: bool = False, pin: Optional[bool] = None, device=None):
        if batch_size < 1 or world_size < 1 or not (0 <= rank < world_size):
            raise ValueError("bad batch_size / rank / world_size")
        self.ds, self.batch_size, self.rank, self.world = dataset, int(batch_size), int(rank), int(world_size)
        self.seed, self.shuffle, self.with_codes = int(seed), bool(shuffle), bool(with_codes)
        self.epoch = 0
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)
        self.device = None if device is None else torch.device(device)
        self._staging = [None, None]
        self._copied = [None, None]  # per staging set: event behind its last host-to-device copies
        self._flip = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return (len(self.ds) // self.batch_size) // self.world

    def _order(self) -> np.ndarray:
        n = len(self.ds)
        if not self.shuffle:
            return np.arange(n, dtype=np.int64)
        return np.random.default_rng([self.seed, self.epoch]).permutation(n).astype(np.int64)

    def batch_indices(self) -> Iterator[np.ndarray]:
        order = self._order()
        for i in range(len(self)):
            b = i * self.world + self.rank
            yield order[b * self.batch_size:(b + 1) * self.batch_size]

    def _buffers(self) -> Tuple[int, Dict[str, torch.Tensor]]:
        slot = self._flip
        self._flip ^= 1
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()  # the copies queued from this set two batches ago have read it
            self._copied[slot] = None
        if self._staging[slot] is None:
            names = ["q_emb", "z_content", "z_style", "genre_idx"] + (["codes"] if self.with_codes else [])
            if self.ds.mert_feat is not None:
                names.append("mert_feat")
            bufs = {}
            for k in names:
                src = getattr(self.ds, k)
                t = torch.empty((self.batch_size,) + tuple(src.shape[1:]), dtype=torch.from_numpy(src[:0]).dtype)
                bufs[k] = t.pin_memory() if self.pin else t
            self._staging[slot] = bufs
        return slot, self._staging[slot]

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        for idx in self.batch_indices():
            slot, bufs = self._buffers()
            sel = np.sort(idx)  # ascending rows: a cheaper gather; the batch is a set, its internal order carries nothing
            for k, t in bufs.items():
                np.take(getattr(self.ds, k), sel, axis=0, out=t.numpy())
            if self.device is not None and self.device.type == "cuda":
                # labels stay on the host (train_codec_stage samples targets from them row by row: a device tensor there
                # would cost a stream-draining read-back per batch)
                batch = {k: (t.clone() if k == "genre_idx" else t.to(self.device, non_blocking=True)) for k, t in bufs.items()}
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._copied[slot] = ev
            else:
                batch = {k: t.clone() for k, t in bufs.items()}  # host consumer: hand out copies, the set is reused
            if "codes" in batch:
                batch["codes"] = batch["codes"].long()
            yield batch
        self.epoch += 1


__all__ = ["CachedCodecDataset", "CodecCacheMeta", "ShardedCodecLoader", "build_genre_exemplar_index",
           "load_codec_cache", "save_codec_cache"]
