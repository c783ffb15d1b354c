"""Worker of tests/test_gpu_dp_two_ranks.py: rank `RANK` of a two-rank gloo group, BOTH ranks on cuda:0 (RCCL refuses two ranks on one
device; gloo moves CUDA tensors through the host).  Each rank runs codec_train_iteration on its half of a global batch of four under
gan_amd.dp.DataParallel; rank 0 then runs the same two iterations on the whole batch in one process and writes the comparison.
usage: dp_two_rank_worker.py <after|overlap> <out.json>   (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment)"""
import datetime
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import lab3_codec_train as T
from gan_amd import ops
from gan_amd.dp import DataParallel
from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.optim import FusedAdamW


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))  # a stuck collective raises
    t_start = time.time()
    note = lambda m: print(f"[rank {rank} +{time.time() - t_start:6.2f}s] {m}", file=sys.stderr, flush=True)
    note("group up")
    dev = torch.device("cuda", 0)
    ops.LSTM_PERSIST = False  # two processes on one GPU: two persistent grids could not both be resident (fp32 mode does not use it anyway)
    torch.manual_seed(45)
    G0 = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="fp32").to(dev).train()
    D0 = MultiScaleWaveDiscriminator(2, precision="fp32").to(dev).train()
    note("models built")
    codec = FrozenEncodec(random_init=True, device="cuda", precision="fp32")
    note("codec built")
    Bg, Tn = 4, 12
    q = 0.5 * torch.randn(Bg, 128, Tn, device=dev)
    zc, zs = torch.randn(Bg, 128, device=dev), torch.randn(Bg, 128, device=dev)
    nz, nz2 = torch.randn(Bg, 32, device=dev), torch.randn(Bg, 32, device=dev)
    qr, qg = 0.5 * torch.randn(Bg, 128, Tn, device=dev), 0.5 * torch.randn(Bg, 128, Tn, device=dev)
    # every term of the north-star subset that is a mean over the batch, plus MR-STFT (whose spectral-convergence ratio is made exact over
    # the global batch by dp.stft_sum_hook); the delta-budget hinge is relu(batch mean - budget), not a mean of per-sample terms: off
    cfg = T.CodecStageTrainConfig(stage_name="stage3", epochs=1, d_step_period=1, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
                                  weights=CodecTrainWeights(content=0.0, style=0.0, delta_budget=0.0, mrstft=float(os.environ.get("GA_DP2_MRSTFT", "2.0"))))

    def run(sl, dp):
        G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="fp32").to(dev).train()
        D = MultiScaleWaveDiscriminator(2, precision="fp32").to(dev).train()
        G.load_state_dict(G0.state_dict())
        D.load_state_dict(D0.state_dict())
        og, od = FusedAdamW(G.parameters(), lr=1e-3, betas=cfg.betas), FusedAdamW(D.parameters(), lr=1e-3, betas=cfg.betas)
        if dp is not None:
            dp.broadcast_state(0, og, od)
            note("state broadcast")
        first = None
        for step in (1, 2):
            note(f"iteration {step} ({'dp' if dp is not None else 'single'})")
            out = T.codec_train_iteration(cfg, G, D, codec, og, od, q[sl], zc[sl], zs[sl], nz[sl], qr[sl], qg[sl], global_step=step,
                                          noise2=nz2[sl], dp=dp)
            if first is None:
                first = (og.flat_g.clone(), od.flat_g.clone(), og.grad_scale, od.grad_scale, float(out["loss_g"]), float(out["loss_d"]))
        torch.cuda.synchronize()
        return og.flat_p.clone(), od.flat_p.clone(), first

    dp = DataParallel(bucket_bytes=256 << 10)
    dp.overlap = mode == "overlap"
    if os.environ.get("GA_DP2_TRACE"):  # where a rank is when its peer gives up
        for name in ("stft_sum_hook", "allreduce_flat", "finish_backward"):
            def traced(*a, _f=getattr(dp, name), _n=name, **k):
                torch.cuda.synchronize()
                note(f"-> {_n}")
                r = _f(*a, **k)
                torch.cuda.synchronize()
                note(f"<- {_n}")
                return r
            setattr(dp, name, traced)
    half = Bg // world
    run(slice(0, 1), None)  # every kernel loaded and every lazy initialisation done before the first collective with a deadline
    note("warm-up iterations done")
    dist.barrier()
    pg, pd, first = run(slice(rank * half, (rank + 1) * half), dp)
    note("two data-parallel iterations done")
    # the ranks hold the same parameters afterwards
    mine = torch.cat([pg, pd]).cpu()  # (gloo moves CUDA tensors for broadcast and all_reduce only: gather on the host)
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    same_across_ranks = bool(torch.equal(both[0], both[1]))
    losses = torch.tensor([first[4], first[5]], dtype=torch.float64)
    dist.all_reduce(losses)
    losses /= world
    dist.barrier()
    if rank == 0:
        rg, rd, rfirst = run(slice(0, Bg), None)
        note("single-process iterations done")
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        l2 = lambda a, b: float((a - b).norm() / b.norm())
        dg, dd = first[0] * first[2] - rfirst[0] * rfirst[2], first[1] * first[3] - rfirst[1] * rfirst[3]
        names_g = [n for n, _ in G0.named_parameters()]
        names_d = [n for n, _ in D0.named_parameters()]
        # the same two shards WITHOUT data parallelism, averaged by hand: the discriminator's gradients (taken before any update) must be
        # what the all-reduce produced; against the whole batch in one process the kernels' batch-dependent summation order shows
        a, b = run(slice(0, half), None)[2], run(slice(half, Bg), None)[2]
        d_vs_shards = l2(first[1] * first[3], 0.5 * (a[1] + b[1]))
        note(f"D gradient: all-reduced vs hand-averaged shards {d_vs_shards:.3e}; hand-averaged shards vs whole batch {l2(0.5 * (a[1] + b[1]), rfirst[1]):.3e}")
        note(f"grad l2: G {l2(first[0] * first[2], rfirst[0] * rfirst[2]):.3e}  D {l2(first[1] * first[3], rfirst[1] * rfirst[3]):.3e}")
        res = {"mode": mode, "same_across_ranks": same_across_ranks, "argmax_g": int(dg.abs().argmax()), "argmax_d": int(dd.abs().argmax()),
               "grad_d_vs_hand_averaged_shards": d_vs_shards, "numel_g": int(dg.numel()), "numel_d": int(dd.numel()), "n_params_g": len(names_g), "n_params_d": len(names_d),
               "grad_g_step1": rel(first[0] * first[2], rfirst[0] * rfirst[2]), "grad_d_step1": rel(first[1] * first[3], rfirst[1] * rfirst[3]),
               "param_g_step2": rel(pg, rg), "param_d_step2": rel(pd, rd),
               "param_g_l2": float((pg - rg).norm() / rg.norm()), "param_d_l2": float((pd - rd).norm() / rd.norm()),
               "loss_g_mean_of_ranks": float(losses[0]), "loss_g_single": rfirst[4], "loss_d_mean_of_ranks": float(losses[1]), "loss_d_single": rfirst[5]}
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
