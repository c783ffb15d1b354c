"""GPU, one-rank RCCL group: the data-parallel code path (gan_amd/dp.py) inside codec_train_iteration gives the parameters
of the plain path - all-reduce after the backward pass, and all-reduce overlapped with it (buckets sent from autograd hooks
and from the deferred weight-gradient sink as their layers complete) - including an R1 step (gradients already in p.grad)
and mode seeking (the generator runs twice).  A one-rank SUM all-reduce is the identity, so the bound is fp32 summation
order.  The 2-rank arithmetic is covered on gloo (tests/test_dp_gloo_cpu.py); no multi-GPU hardware result exists yet."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from gan_amd import lab3_codec_train as T
from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.optim import FusedAdamW

DEV = torch.device("cuda")


@pytest.fixture(scope="module")
def one_rank_group():
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        yield
        dist.destroy_process_group()
    else:
        yield


@pytest.mark.parametrize("mode_seeking", [0.0, 0.7], ids=["plain", "mode_seeking"])
def test_dp_paths_equal_the_single_process_path(one_rank_group, mode_seeking):
    from gan_amd.dp import DataParallel
    torch.manual_seed(45)
    G0 = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="fp32").to(DEV).train()
    D0 = MultiScaleWaveDiscriminator(2, precision="fp32").to(DEV).train()
    codec = FrozenEncodec(random_init=True, device="cuda", precision="fp32")  # fp32 mode: the comparison is about the DP plumbing
    B, Tn = 2, 12
    q = 0.5 * torch.randn(B, 128, Tn, device=DEV)
    zc, zs, nz = torch.randn(B, 128, device=DEV), torch.randn(B, 128, device=DEV), torch.randn(B, 32, device=DEV)
    nz2 = torch.randn(B, 32, device=DEV)
    qr, qg = 0.5 * torch.randn(B, 128, Tn, device=DEV), 0.5 * torch.randn(B, 128, Tn, device=DEV)
    cfg = T.CodecStageTrainConfig(stage_name="stage3", epochs=1, d_step_period=1, r1_gamma=2.0, r1_interval=1,
                                  g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
                                  weights=CodecTrainWeights(content=0.0, style=0.0, mode_seeking=mode_seeking))
    results = {}
    for name in ("none", "after", "overlap"):
        G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="fp32").to(DEV).train()
        D = MultiScaleWaveDiscriminator(2, precision="fp32").to(DEV).train()
        G.load_state_dict(G0.state_dict())
        D.load_state_dict(D0.state_dict())
        og, od = FusedAdamW(G.parameters(), lr=1e-3, betas=cfg.betas), FusedAdamW(D.parameters(), lr=1e-3, betas=cfg.betas)
        dp = None
        if name != "none":
            dp = DataParallel(bucket_bytes=256 << 10)  # small buckets: several per network, weight / gain pairs split across them
            assert dp.overlap is False                 # one rank: nothing to hide, so the default stays "after the pass"
            dp.overlap = name == "overlap"
            dp.broadcast_state(0, og, od)
            assert len(dp.bucket_slices_for(od)) > 3
        grads, losses = [], []
        for step in (1, 2):  # every step is an R1 step (gradients already in p.grad before the hinge backward)
            out = T.codec_train_iteration(cfg, G, D, codec, og, od, q, zc, zs, nz, qr, qg, global_step=step, noise2=nz2, dp=dp)
            grads.append((og.flat_g.clone(), od.flat_g.clone()))
            losses.append((float(out["loss_g"]), float(out["loss_d"])))
        torch.cuda.synchronize()
        results[name] = (og.flat_p.clone(), od.flat_p.clone(), grads, losses[0][0], losses[0][1], losses[1])
    ref = results["none"]
    for name in ("after", "overlap"):
        got = results[name]
        assert got[3] == pytest.approx(ref[3], rel=1e-5) and got[4] == pytest.approx(ref[4], rel=1e-5), name
        for (gg, gd), (rg, rd) in zip(got[2][:1], ref[2][:1]):  # first step: identical inputs -> gradients equal to summation order
            for a, b, what in ((gg, rg, "G grad"), (gd, rd, "D grad")):
                err = float((a - b).abs().max() / b.abs().max())
                assert err <= 1e-5, (name, what, err)
        # two AdamW steps later (the first ones are ~ lr * sign(g): 1e-7 differences in g show up at 1e-4 of a parameter)
        for a, b, what in ((got[0], ref[0], "G"), (got[1], ref[1], "D")):
            d = (a - b).abs() / b.abs().max()
            l2 = float((a - b).norm() / b.norm())
            assert float(d.max()) <= 5e-3 and l2 <= 3e-3, (name, what, float(d.max()), l2)
        assert got[5][0] == pytest.approx(ref[5][0], rel=2e-3) and got[5][1] == pytest.approx(ref[5][1], rel=2e-3), name


def test_bucket_order_follows_the_backward_pass(one_rank_group):
    """Buckets are closed at parameter boundaries, walking from the LAST parameter: the first bucket holds the tail of the last
    discriminator scale (the gradients the backward pass produces first), SURVEY.md 8(e)."""
    from gan_amd.dp import DataParallel
    D = MultiScaleWaveDiscriminator(3, precision="bf16").to(DEV)
    od = FusedAdamW(D.parameters(), lr=1e-3)
    dp = DataParallel()  # default bucket size
    sl = dp.bucket_slices_for(od)
    assert sl[0][1] == od.numel and sl[-1][0] == 0 and all(a < b for a, b in sl)
    assert [b for _, b in sl[1:]] == [a for a, _ in sl[:-1]]           # contiguous, descending
    assert all(a in od.offsets or a == 0 for a, _ in sl)               # closed at parameter boundaries
    names = [k for k, _ in D.named_parameters()]
    first = [n for n, off in zip(names, od.offsets) if off >= sl[0][0]]
    assert first[-1].startswith("discs.2.out") and any(n.startswith("discs.2.blocks.4") for n in first)
    assert 3 <= len(sl) <= 5                                           # 85.7 MB of D gradients in 24 MiB buckets


def test_bench_legs_run_through_under_overlapped_data_parallelism():
    """Every leg of bench.py under a one-rank RCCL group with the all-reduce overlapped with the backward pass, in a process of its own
    with a time limit: with a prioritised side stream this very sequence (fp32 leg, then the stage-2-like leg) deadlocked -- more streams
    than hardware queues -- which is why gan_amd.lab3_codec_train.SIDE_STREAM_PRIORITY is 0."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GA_SIDE_PRIO="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):  # outside torchrun: bench.py opens its own one-rank group
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--force-dp", "--dp-overlap", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["dp_allreduce"].startswith("overlapped")
    for leg in ("batch12", "batch128", "stage2_like", "fp32_batch12"):
        assert line[leg]["value"] > 0
