"""Worker of tests/test_gpu_dp_two_ranks.py::test_mel_stage1_*: rank `RANK` of a two-rank gloo group, both ranks on cuda:0, the mel
track's stage-1 iteration under gan_amd.dp.DataParallel on half of a global batch of four each.  The discriminator has BatchNorm, whose
statistics are per rank (DDP without SyncBN), so the reference point is not the whole batch in one process but the two shards run
WITHOUT data parallelism and averaged by hand; the MR-STFT term, which does not pass through BatchNorm, is compared with the whole batch.
usage: mel_dp_two_rank_worker.py <out.json>   (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment)"""
import copy
import datetime
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import lab3_models as PM
from gan_amd import lab3_train as PT
from gan_amd.dp import DataParallel
from gan_amd.optim import FusedAdamW


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    G0 = PM.ReconstructionDecoder(n_mels=32, n_frames=64, base_channels=64, precision="fp32").to(dev).train()
    D0 = PM.MelDiscriminator(base_channels=16, cond_dim=160, precision="fp32").to(dev).train()
    Bg = 4
    real = torch.rand(Bg, 32, 64, device=dev) * 2 - 1
    zc, cond = torch.randn(Bg, 128, device=dev), torch.randn(Bg, 160, device=dev)
    full = PT.StageTrainConfig(stage_name="stage1", epochs=1, weights=PT.TrainWeights(content=0.0, style=0.0), mrstft_resolutions=((32, 8, 32), (64, 16, 64)))
    no_stft = PT.StageTrainConfig(stage_name="stage1", epochs=1, d_step_period=2, weights=PT.TrainWeights(content=0.0, style=0.0, continuity=0.0, mrstft=0.0))

    def run(cfg, sl, dp, steps):
        G, D = copy.deepcopy(G0), copy.deepcopy(D0)
        og = FusedAdamW(G.parameters(), lr=cfg.lr_g, betas=cfg.betas, weight_decay=0.0)
        od = FusedAdamW(D.parameters(), lr=cfg.lr_d, betas=cfg.betas, weight_decay=0.0)
        if dp is not None:
            dp.broadcast_state(0, og, od)
        outs = []
        for step in steps:
            out = PT.mel_stage1_iteration(cfg, G, D, og, od, real[sl], zc[sl], cond[sl], step, dp=dp)
            outs.append({k: float(v) for k, v in out.items()})
            if len(outs) == 1:
                first = (og.flat_g.clone() * og.grad_scale, od.flat_g.clone() * od.grad_scale)
        torch.cuda.synchronize()
        return torch.cat([og.flat_p, od.flat_p]).clone(), first, outs

    half = Bg // world
    mine = slice(rank * half, (rank + 1) * half)
    run(full, slice(0, 1), None, (0,))  # kernels loaded before the first collective with a deadline
    dist.barrier()
    dp = DataParallel(bucket_bytes=256 << 10)
    p_a, g_a, o_a = run(full, mine, dp, (0, 1))      # D step + G step, every term; then a second iteration
    p_b, g_b, o_b = run(no_stft, mine, dp, (1,))     # no D step at step 1: the generator's gradient through an un-updated discriminator
    host = p_a.cpu()
    both = [torch.zeros_like(host) for _ in range(world)]
    dist.all_gather(both, host)
    dist.barrier()
    if rank == 0:
        l2 = lambda a, b: float((a - b).norm() / b.norm())
        sh = [run(full, slice(r * half, (r + 1) * half), None, (0,)) for r in range(world)]
        d_avg = sum(s[1][1] for s in sh) / world
        sh2 = [run(no_stft, slice(r * half, (r + 1) * half), None, (1,)) for r in range(world)]
        g_avg = sum(s[1][0] for s in sh2) / world
        whole = run(full, slice(0, Bg), None, (0,))
        res = {"same_across_ranks": bool(torch.equal(both[0], both[1])),
               "grad_d_vs_hand_averaged_shards": l2(g_a[1], d_avg), "grad_g_vs_hand_averaged_shards": l2(g_b[0], g_avg),
               "mrstft_dp": o_a[0]["loss_mrstft"], "mrstft_whole_batch": whole[2][0]["loss_mrstft"],
               "mrstft_rank0_shard_alone": sh[0][2][0]["loss_mrstft"], "l1_dp_rank0": o_a[0]["loss_l1"], "l1_rank0_shard_alone": sh[0][2][0]["loss_l1"]}
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
