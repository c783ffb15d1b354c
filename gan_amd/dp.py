"""Data parallelism for the codec GAN step: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI on ROCm;
"gloo" for the CPU tests).  The reference has no distributed code at all; this is the
MI355X-side design:

  * the batch is sharded across ranks; G, D, decoder and both optimizers are replicated;
  * gradients live in the optimizer's flat fp32 buffer (gan_amd/optim.py) and are SUM-all-reduced in contiguous
    buckets (default 24 MiB, closed at parameter boundaries: xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large messages beat many
    small ones); 1/world_size is folded into the fused AdamW kernel (grad_scale), so no separate scaling pass;
  * buckets are issued asynchronously in reverse parameter order (the order backward produces them) and waited for
    only right before the clip+AdamW kernel;
  * the MR-STFT spectral-convergence term is a ratio of WHOLE-BATCH Frobenius norms (lab3_codec_train.py:170): the
    three per-resolution sums are all-reduced before the ratio so the loss equals the single-process global-batch
    value (the gradient then scales correctly because every rank differentiates the same global ratio).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo takes CUDA tensors for broadcast and all_reduce, but moves them on HIGH-PRIORITY pool streams of its own; on this stack a
    prioritised stream beside more streams than hardware queues has deadlocked twice (lab3_codec_train.SIDE_STREAM_PRIORITY; two gloo
    ranks sharing one GPU, tests/test_gpu_dp_two_ranks.py), so under gloo a device tensor goes through the host explicitly: a blocking
    copy out, the collective on the CPU tensor, a copy back.  Slow and never overlapped -- gloo is the test / fallback backend, RCCL
    ("nccl") is the product's."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _StagedWork:
    """all_reduce of a device tensor through the host (see _host_staged); wait() finishes it."""

    def __init__(self, t: torch.Tensor, group):
        self.t, self.host = t, t.detach().cpu()  # synchronises the current stream with the host
        self.work = dist.all_reduce(self.host, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def wait(self) -> None:
        self.work.wait()
        self.t.copy_(self.host)


def all_reduce_sum(t: torch.Tensor, group, async_op: bool = False):
    if _host_staged(t, group):
        w = _StagedWork(t, group)
        if async_op:
            return w
        w.wait()
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast(t: torch.Tensor, src: int, group) -> None:
    if _host_staged(t, group):
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
        return
    dist.broadcast(t, src=src, group=group)


class _AllReduceSums(torch.autograd.Function):
    """forward: SUM over ranks.  backward: gradient times world_size - every rank then holds d(GLOBAL loss)/d(its own
    shard) * W, and the optimizer's mean over ranks (grad_scale = 1/W) turns the sum of those partial gradients into the
    full-batch gradient, on the same footing as the per-rank-mean loss terms."""

    @staticmethod
    def forward(ctx, t, group):
        out = t.clone()
        all_reduce_sum(out, group)
        ctx.world = dist.get_world_size(group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g * float(ctx.world), None


class _OverlapState:
    """Bucket bookkeeping of one optimizer for DataParallel.begin_backward / finish_backward."""

    def __init__(self, dp: "DataParallel", opt):
        self.dp, self.opt = dp, opt
        self.sink = None  # ops.WeightGradSink of the running backward pass (conv weights' gradients bypass p.grad)
        self.slices = dp.bucket_slices_for(opt)
        self.bucket_of: List[List[int]] = []
        self.need = [0] * len(self.slices)
        for p, off in zip(opt.params, opt.offsets):
            lo, hi = off, off + p.numel()
            bs = [b for b, (a, e) in enumerate(self.slices) if lo < e and hi > a]
            self.bucket_of.append(bs)
            for b in bs:
                self.need[b] += 1
        self.members = [[i for i, bs in enumerate(self.bucket_of) if b in bs] for b in range(len(self.slices))]
        # buckets whose readiness parameter i counts towards: its own, plus - learnt from the sink's offers - those of its
        # weight-norm partner (a layer's deferred gradient is finished for weight and gain together, so a bucket holding
        # one of them must also wait for the other one's contributions)
        self.dep_of: List[List[int]] = [list(bs) for bs in self.bucket_of]
        self.pairs: set = set()
        self.copied: List[bool] = []
        self.cuda = opt.flat_g.is_cuda
        self.comm = torch.cuda.Stream(device=opt.flat_g.device) if self.cuda else None
        self.active = False
        self.left: List[int] = []
        self.sent: List[bool] = []
        self.works: list = []

    def learn_pair(self, iv: int, ig) -> None:
        """sink.offer callback: layer (weight iv, gain ig) exists.  Runs before either parameter's hook of this pass."""
        if ig is None or (iv, ig) in self.pairs:
            return
        self.pairs.add((iv, ig))
        for a, other in ((iv, ig), (ig, iv)):
            for bk in self.bucket_of[other]:
                if bk not in self.dep_of[a]:
                    self.dep_of[a].append(bk)
                    self.need[bk] += 1
                    if self.active:
                        self.left[bk] += 1

    def begin(self, sink=None) -> None:
        self.sink = sink
        if sink is not None:
            sink.on_offer = self.learn_pair
        self.opt.flat_g.zero_()
        self.arrived = [False] * len(self.opt.params)
        self.left = list(self.need)
        self.copied = [False] * len(self.opt.params)
        self.sent = [False] * len(self.slices)
        self.works = []
        self.active = True

    def on_grad(self, i: int) -> None:
        """Post-accumulate-grad hook of parameter i: every contribution of this pass to it has been produced.  (torch runs
        the hook of a parameter whose incoming gradient is undefined too, which is what a conv weight whose gradient went
        to the sink looks like to autograd; should a torch version not do that, such parameters simply wait for finish().)"""
        if not self.active or self.arrived[i]:
            return
        self.arrived[i] = True
        for b in self.dep_of[i]:
            self.left[b] -= 1
            if self.left[b] == 0:
                self._send(b)

    def _send(self, b: int) -> None:
        if self.sent[b]:
            return
        self.sent[b] = True
        # one fused copy of the bucket's gradients into the flat buffer.  A parameter that straddles two buckets is copied
        # with the first of them to go; the other one cannot have gone earlier, because it was waiting for this parameter too.
        # Deferred conv-weight gradients (sink) whose weight or gain lies in this bucket are finished into the flat buffer by
        # one launch; their partner parameter, if it lies in another bucket, is copied now as well so that the finish kernel's
        # accumulate flags (a p.grad that existed before the pass, e.g. R1's, or a second use of the layer in this pass) see
        # the value they add to.  The partner has arrived too: a bucket also waits for the partners of its members (dep_of).
        members = list(self.members[b])
        deferred = []
        if self.sink is not None and self.sink.pending:
            mset = set(members)
            deferred = [iv for iv, ent in self.sink.pending.items() if iv in mset or (ent[0] is not None and ent[0] in mset)]
            for iv in deferred:
                ig = self.sink.pending[iv][0]
                members += [i for i in (iv, ig) if i is not None and i not in mset]
        src, dst = [], []
        for i in members:
            if not self.copied[i]:
                self.copied[i] = True
                g = self.opt.params[i].grad
                if g is not None and g.data_ptr() != self.opt.grad_views[i].data_ptr():
                    src.append(g)
                    dst.append(self.opt.grad_views[i])
        if src:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)
        if deferred:
            self.sink.flush(only=deferred)
        a, e = self.slices[b]
        chunk = self.opt.flat_g[a:e]
        if self.cuda and not _host_staged(chunk, self.dp.group):
            ev = torch.cuda.Event()
            ev.record()  # the copies into this bucket are on the current (main) stream
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.dp.group, async_op=True))
        else:
            self.works.append(all_reduce_sum(chunk, self.dp.group, async_op=True))

    def finish(self) -> None:
        self.active = False
        for b in range(len(self.slices)):
            self._send(b)
        if self.sink is not None:
            self.sink.on_offer = None
        self.sink = None
        for w in self.works:
            w.wait()  # (RCCL: the current stream waits; no host synchronisation)
        self.works = []


class DataParallel:
    def __init__(self, group=None, bucket_bytes: int = 24 << 20, exact_stft: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.bucket_elems = max(1, int(bucket_bytes) // 4)
        self.exact_stft = exact_stft
        # True: all-reduce each bucket while the backward pass is still running (begin/finish_backward; north star: "all-reduce
        # overlapped with the discriminator backward pass").  OPT-IN (`DataParallel.overlap = True`, `bench.py --dp-overlap`): the path
        # has never run on more than one GPU (no multi-GPU node in rounds 1-3: SCALE_r0*.json skipped) -- it rests on the gloo
        # world-2 CPU tests and the forced one-rank RCCL tests -- and its extra streams have already produced one deadlock on a
        # single GPU (a prioritised side stream beside the communication streams, lab3_codec_train.SIDE_STREAM_PRIORITY).  What runs
        # concurrently on hardware is RCCL's kernels beside the persistent LSTM backward, which needs one workgroup resident on
        # every CU; its bounded waits turn a lost residency into an error word, and check_first_overlapped_iteration() below reads it
        # after the first overlapped iteration.  The default, all-reduce after the backward pass, costs the un-hidden collective:
        # 135 MB of gradients per step, ~2-3 ms over xGMI at 8 GPUs against a 47 ms step.
        self.overlap = False
        self._overlap_checked = False

    def check_first_overlapped_iteration(self, device) -> None:
        """Once, after the first iteration that overlapped collectives with the backward pass: a persistent LSTM launch that lost
        its residency to a co-running RCCL kernel has raised the sticky error word (gan_amd.ops.lstm_persist_error)."""
        if self._overlap_checked or not self.overlap:
            return
        self._overlap_checked = True
        from . import ops
        if torch.device(device).type == "cuda" and ops.lstm_persist_error(device):
            raise RuntimeError("a persistent LSTM launch ran out of its bounded wait while gradient all-reduces were overlapped with "
                               "the backward pass: set DataParallel.overlap = False (bench.py --no-dp-overlap) or "
                               "gan_amd.ops.LSTM_PERSIST = False")

    def reduce_history_means(self, host: dict, counts: dict) -> None:
        """Per-epoch loss sums and their iteration counts summed over the ranks, in place (the history row is then the mean over
        the GLOBAL batch, the same on every rank, instead of rank 0's shard only)."""
        if self.world_size == 1:
            return
        keys = sorted(host)
        ckeys = sorted(counts)
        t = torch.tensor([float(host[k]) for k in keys] + [float(counts[k]) for k in ckeys], dtype=torch.float64)
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        vals = t.cpu().tolist()
        for k, v in zip(keys, vals[:len(keys)]):
            host[k] = v
        for k, v in zip(ckeys, vals[len(keys):]):
            counts[k] = v

    def bucket_slices(self, numel: int) -> List[Tuple[int, int]]:
        """Contiguous [start, end) slices of the flat gradient buffer, last parameters first."""
        out = []
        end = numel
        while end > 0:
            start = max(0, end - self.bucket_elems)
            out.append((start, end))
            end = start
        return out

    def bucket_slices_for(self, opt) -> List[Tuple[int, int]]:
        """Buckets of one optimizer's flat gradient buffer in the order the backward pass completes them: parameters are
        walked from the LAST one (D: scale 2 ... scale 0, each `out` -> blocks.4 -> ... -> blocks.0) and a
        bucket is closed at a parameter boundary once it holds bucket_elems, so no tensor straddles two collectives."""
        params, offsets = getattr(opt, "params", None), getattr(opt, "offsets", None)
        if not params:
            return self.bucket_slices(opt.numel)
        out, end = [], opt.numel
        for i in range(len(params) - 1, -1, -1):
            if end - offsets[i] >= self.bucket_elems or i == 0:
                out.append((0 if i == 0 else offsets[i], end))
                end = offsets[i]
        return out

    def sync_sqnorm(self, sq: torch.Tensor) -> None:
        """The clip coefficient has to be the SAME NUMBER on every rank.  After the all-reduce the ranks hold bit-identical gradients,
        but sum(g^2) is taken with floating-point atomics whose order differs from launch to launch, so each rank would clip by its
        own last bits and the replicas would drift apart (found by tests/test_gpu_dp_two_ranks.py: parameters equal across ranks in
        one run, unequal in the next).  MAX over ranks of the one-float device scalar, on the stream, between the norm kernel and
        the AdamW kernel (installed as opt.sqnorm_hook)."""
        if self.world_size == 1:
            return
        if _host_staged(sq, self.group):
            host = sq.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            sq.copy_(host)
        else:
            dist.all_reduce(sq, op=dist.ReduceOp.MAX, group=self.group)

    def allreduce_flat(self, opt) -> None:
        """SUM-all-reduce opt.flat_g bucket by bucket; the mean is applied inside the AdamW kernel."""
        opt.sqnorm_hook = self.sync_sqnorm
        works = []
        for a, b in self.bucket_slices_for(opt):
            works.append(all_reduce_sum(opt.flat_g[a:b], self.group, async_op=True))
        for w in works:
            w.wait()
        opt.grad_scale = 1.0 / float(self.world_size)

    # ---- all-reduce overlapped with the backward pass --------------------------------------------------------------
    # Every parameter gets a post-accumulate-grad hook that copies its finished gradient into the optimizer's flat buffer
    # and counts it off against the bucket(s) it lies in; the moment a bucket is complete its SUM all-reduce is issued on a
    # side stream, while autograd keeps producing the earlier layers' gradients on the main stream (backward yields the
    # LAST parameters first, so the highest bucket -- D blocks.4, most of the bytes -- goes out first).  finish_backward
    # issues whatever is left (parameters that received no gradient stay zero) and makes the main stream wait.
    def attach(self, opt) -> None:
        if getattr(opt, "_dp_overlap", None) is not None:
            return
        st = _OverlapState(self, opt)
        opt._dp_overlap = st
        for i, p in enumerate(opt.params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i, st=st: st.on_grad(i))

    def begin_backward(self, opt, sink=None) -> None:
        self.attach(opt)
        opt._dp_overlap.begin(sink)

    def finish_backward(self, opt) -> None:
        opt._dp_overlap.finish()
        opt.sqnorm_hook = self.sync_sqnorm
        opt.grad_scale = 1.0 / float(self.world_size)

    def stft_sum_hook(self, sums: torch.Tensor, count: float):
        if not self.exact_stft or self.world_size == 1:
            return sums, count
        g = _AllReduceSums.apply(sums, self.group)
        return g, count * self.world_size

    def broadcast_state(self, start_epoch: int, *opts) -> int:
        """Rank 0's parameters, AdamW moments, step counters and resume epoch to every rank (train_codec_stage calls this once
        after the optimizers are built / a checkpoint is loaded).  Returns rank 0's start epoch."""
        self.broadcast_params(*opts)
        meta = torch.tensor([float(start_epoch)] + [float(o.step_count) for o in opts], dtype=torch.float64,
                            device=opts[0].flat_p.device)
        broadcast(meta, 0, self.group)
        for opt in opts:
            broadcast(opt.exp_avg, 0, self.group)
            broadcast(opt.exp_avg_sq, 0, self.group)
        vals = meta.tolist()
        for o, v in zip(opts, vals[1:]):
            o.step_count = int(v)
        return int(vals[0])

    def broadcast_params(self, *opts) -> None:
        for opt in opts:
            broadcast(opt.flat_p, 0, self.group)
            epoch = getattr(opt, "_epoch", None)
            if epoch is not None:
                epoch[0] += 1  # parameters rewritten in place: cached weight packs are stale
