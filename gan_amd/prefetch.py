"""Side-stream decode of the NEXT batch's frozen latents.

The frozen decodes of an iteration (x_src, x_real for the D step, x_real for the G step: reference
lab3_codec_train.py:415,422,458) depend on nothing the iteration computes: the decoder never changes and the latents come
straight from the loader and the exemplar bank.  Their LSTM is a chain of ~376 dependent, latency-bound launches that
leaves most of the GPU idle, and so are the generator's small convolutions.  Running the next batch's frozen decode on
a second HIP stream while the current iteration runs on the main stream lets the two latency-bound chains share the chip.
The reference has no counterpart (it is single-stream PyTorch); results are unchanged.

Status: OPTIONAL and off by default.  Measured on MI355X (ROCm 7.2, scratch/bench_streams.py): a B=12 and a B=36 decode
take 3.1 + 5.8 ms back to back and 8.2 ms on two streams - the dependent step launches of the two chains barely
overlap - and the full training step is 0.9 ms slower with the prefetch than with the single batched LSTM chain.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


class FrozenDecodePrefetcher:
    def __init__(self, codec):
        self.codec = codec
        self.stream: Optional[torch.cuda.Stream] = None
        self._pending = None  # (key, out, done_event, inputs)

    @staticmethod
    def _key(qs: Sequence[torch.Tensor]) -> Tuple:
        return tuple((int(t.data_ptr()), tuple(t.shape), t._version) for t in qs)

    def launch(self, qs: List[torch.Tensor]) -> None:
        """Start decoding cat(qs) on the side stream.  The tensors must not be modified until take() returned."""
        if not qs:
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=qs[0].device)
        main = torch.cuda.current_stream(qs[0].device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            with torch.no_grad():
                out = self.codec.decode_embeddings(torch.cat(list(qs), dim=0))
            done = torch.cuda.Event()
            done.record(self.stream)
        for t in qs:
            t.record_stream(self.stream)
        self._pending = (self._key(qs), out, done, list(qs))

    def take(self, qs: List[torch.Tensor]) -> Optional[torch.Tensor]:
        """The decode launched for exactly these tensors, or None (caller decodes inline)."""
        pend, self._pending = self._pending, None
        if pend is None or pend[0] != self._key(qs):
            return None
        _, out, done, _inputs = pend
        main = torch.cuda.current_stream(out.device)
        main.wait_event(done)
        out.record_stream(main)
        return out
