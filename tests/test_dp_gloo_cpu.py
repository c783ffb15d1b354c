"""CPU, world_size 2, gloo: the data-parallel host logic (gan_amd/dp.py) - bucketed SUM all-reduce of the flat gradient
buffer with the mean folded into grad_scale, parameter broadcast, and the exact global-batch MR-STFT ratio."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeOpt:
    """The two attributes DataParallel touches on gan_amd.optim.FusedAdamW."""

    def __init__(self, n):
        self.numel = n
        self.flat_g = torch.zeros(n)
        self.flat_p = torch.zeros(n)
        self.grad_scale = 1.0


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gan_amd.dp import DataParallel
        from oracle import codec_oracle as O
        dp = DataParallel(bucket_bytes=4096)  # 1024-element buckets -> several buckets, last one ragged
        n = 5000
        slices = dp.bucket_slices(n)
        assert slices[0][1] == n and slices[-1][0] == 0 and all(a < b for a, b in slices)
        assert sum(b - a for a, b in slices) == n and len(slices) == 5
        opt = _FakeOpt(n)
        opt.flat_g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        dp.allreduce_flat(opt)
        assert torch.equal(opt.flat_g, torch.arange(n, dtype=torch.float32) * 3)  # 1x + 2x
        assert opt.grad_scale == 0.5
        opt.flat_p = torch.full((n,), float(rank + 7))
        dp.broadcast_params(opt)
        assert torch.equal(opt.flat_p, torch.full((n,), 7.0))

        # all-reduce overlapped with backward: hooks copy each finished gradient into the flat buffer and send a bucket as
        # soon as it is complete; a parameter straddling two buckets and one that receives no gradient are both covered
        class _Opt2:
            pass
        torch.manual_seed(3)
        o2 = _Opt2()
        o2.params = [torch.nn.Parameter(torch.randn(700)), torch.nn.Parameter(torch.randn(30, 50)),
                     torch.nn.Parameter(torch.randn(9)), torch.nn.Parameter(torch.randn(1200))]
        o2.offsets, tot = [], 0
        for prm in o2.params:
            o2.offsets.append(tot)
            tot += (prm.numel() + 3) // 4 * 4
        o2.numel = tot
        o2.flat_g = torch.full((tot,), 5.0)  # stale contents must not survive
        o2.grad_views = [o2.flat_g[off:off + prm.numel()].view_as(prm) for prm, off in zip(o2.params, o2.offsets)]
        o2.grad_scale = 1.0
        for rep in range(2):  # the hooks stay registered: a second backward must start clean
            for prm in o2.params:
                prm.grad = None
            loss = (o2.params[0] * (rank + 1.0)).sum() + (o2.params[1] ** 2).sum() * (rank + 2.0) + (o2.params[3] * 0.5).sum()
            dp.begin_backward(o2)
            loss.backward()
            dp.finish_backward(o2)
            assert o2.grad_scale == 0.5
            want = [torch.full((700,), 3.0), 2.0 * o2.params[1].detach() * 5.0, torch.zeros(9), torch.full((1200,), 1.0)]
            for v, wv in zip(o2.grad_views, want):
                assert torch.allclose(v, wv.view_as(v), rtol=1e-6), rep
        # outside begin/finish the hooks are inert
        o2.flat_g.fill_(-1.0)
        (o2.params[0].sum()).backward()
        assert torch.equal(o2.flat_g, torch.full((tot,), -1.0))

        # exact global-batch spectral convergence: ratio of all-reduced sums == single-process value on the full batch
        torch.manual_seed(0)
        x = torch.randn(4, 4096)
        y = torch.randn(4, 4096)
        n_fft, hop = 512, 128
        mx = O.stft_mag(x, n_fft, hop, n_fft).clamp_min(1e-8)
        my = O.stft_mag(y, n_fft, hop, n_fft).clamp_min(1e-8)
        want_sc = torch.sqrt(((my - mx) ** 2).sum()) / (torch.sqrt((my ** 2).sum()) + 1e-8)
        want_lm = (mx.log() - my.log()).abs().mean()
        sl = slice(2 * rank, 2 * rank + 2)
        lx = mx[sl].clone().requires_grad_(True)
        sums = torch.stack([((my[sl] - lx) ** 2).sum(), (my[sl] ** 2).sum(), (lx.log() - my[sl].log()).abs().sum()])
        gsums, count = dp.stft_sum_hook(sums, float(lx.numel()))
        sc = gsums[0].sqrt() / (gsums[1].sqrt() + 1e-8)
        lm = gsums[2] / count
        assert float(sc) == pytest.approx(float(want_sc), rel=1e-5)
        assert float(lm) == pytest.approx(float(want_lm), rel=1e-5)
        # gradient: every rank differentiates the global loss w.r.t. its own shard, times world_size, because the
        # optimizer afterwards averages the flat gradients over ranks (grad_scale = 1/W).
        (sc + lm).backward()
        fx = mx.clone().requires_grad_(True)
        full = torch.sqrt(((my - fx) ** 2).sum()) / (torch.sqrt((my ** 2).sum()) + 1e-8) + (fx.log() - my.log()).abs().mean()
        full.backward()
        assert torch.allclose(lx.grad, world * fx.grad[sl], rtol=1e-4, atol=1e-8)
        # per-epoch history: loss sums and iteration counts are summed over the ranks, so the row is the global-batch mean
        host, counts = {"loss_g": 3.0 * (rank + 1), "loss_d": 1.0 + rank}, {"n": 4.0, "d": 2.0}
        dp.reduce_history_means(host, counts)
        assert host == {"loss_g": 9.0, "loss_d": 3.0} and counts == {"n": 8.0, "d": 4.0}
        dp.check_first_overlapped_iteration("cpu")  # (CPU: nothing to read; must be a no-op the second time too)
        dp.check_first_overlapped_iteration("cpu")
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, "ok"), (1, "ok")], res
    assert all(p.exitcode == 0 for p in procs)
