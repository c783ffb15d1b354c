#!/usr/bin/env python3
"""Benchmark of the lab3 codec-track G+D training step on MI355X (BASELINE.json metric: train samples/s).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of train_codec_stage's loop body (reference lab3_codec_train.py:404-591) on one
synthetic batch of the full codec shape: d_step_period=1, no R1, loss terms {adv, FM, latent L1, continuity, MR-STFT,
delta budget}, clip 5.0 + AdamW on both networks - SURVEY.md section 8(d)'s unit of work.  Inputs are resident in HBM
before the timed region.  Weak scaling: every rank processes its own batch of `--batch` samples; gradients are
all-reduced over RCCL (gan_amd/dp.py).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_NAMES = {0: "conv_direct_kernel", 1: "conv_fewout_kernel", 2: "conv_cin1_kernel",
                3: "conv_mfma_kernel<1,4,32>", 4: "conv_mfma_kernel<2,2,64>", 5: "conv_fewout_rows_kernel",
                6: "conv_cin1_vec_kernel", 7: "conv_gemm_kernel<2,1,4>", 8: "conv_gemm_kernel<2,2,2>", 9: "conv_gemm_kernel<1,1,4>", 10: "conv_stream_kernel", 11: "conv_pp_kernel", 12: "conv_win_kernel", 13: "conv_swin_kernel", 14: "resunit_fwd_kernel", 15: "resunit_bwd_kernel", 16: "conv_coop_kernel", 17: "conv_dual_k1_kernel"}


def kernel_name(kid: int) -> str:
    """Kernel ids of the convolution launches (include/gan_amd.h GA_KERNEL_*); the gathered-GEMM instantiations arrive as
    0xMWNB | BUF << 16 (ga_conv_gemm_variant) and are named as a profiler names them."""
    if kid >= 0x1000:
        mi, wm, wn, nb = (kid >> 12) & 15, (kid >> 8) & 15, (kid >> 4) & 15, kid & 15
        return f"conv_gemm_kernel<{mi},{wm},{wn},{nb},{'true' if kid & 0x10000 else 'false'}>"
    return KERNEL_NAMES.get(kid, f"kernel_{kid}")


def make_models(precision: str, device, seed: int):
    from gan_amd.lab3_codec_bridge import FrozenEncodec
    from gan_amd.lab3_codec_models import CodecLatentTranslator, MultiScaleWaveDiscriminator
    torch.manual_seed(seed)
    G = CodecLatentTranslator(precision=precision).to(device).train()
    D = MultiScaleWaveDiscriminator(3, precision=precision).to(device).train()
    codec = FrozenEncodec(random_init=True, device=device, precision=precision)
    return G, D, codec


def make_batch(B: int, T: int, device, gen: torch.Generator):
    import torch.nn.functional as F
    r = lambda *s: torch.randn(*s, generator=gen)
    q_src = 0.5 * r(B, 128, T)
    zc = F.normalize(r(B, 128), dim=-1)
    zs = F.normalize(r(B, 128), dim=-1)
    noise = r(B, 32)
    bank = 0.5 * r(4 * 64, 128, T)  # 4 genres x 64 exemplars
    idx_d = torch.randint(0, bank.shape[0], (B,), generator=gen)
    idx_g = torch.randint(0, bank.shape[0], (B,), generator=gen)
    return [t.to(device) for t in (q_src, zc, zs, noise, bank[idx_d], bank[idx_g])]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T: int, batches=(4, 12), iters: int = 5, budget_s: float = 75.0):
    """The CPU oracle (oracle/codec_oracle.py, a torch-CPU port of the reference step pinned against the reference's own
    outputs) timed on the host cores, SURVEY.md 8(d): B = 4 and B = 12, one warm-up + up to `iters` iterations each, median,
    all the cores this job may use, CPU model stated.  Bounded: a leg stops taking iterations once `budget_s` seconds are
    spent (at least two timed iterations), and says how many it took."""
    import statistics
    from oracle import codec_oracle as O
    n_threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(n_threads)
    cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0))
    t_start = time.perf_counter()
    legs = {}
    for bi, B in enumerate(batches):
        g_sd, d_sd, dec_sd = O.fresh_state()
        gen = torch.Generator().manual_seed(328)
        og, od = O.AdamWState(), O.AdamWState()
        batch = [t.cpu() for t in make_batch(B, T, "cpu", gen)]
        O.iteration(g_sd, d_sd, dec_sd, og, od, batch, 1, cfg)  # warm-up
        times = []
        share = budget_s * (bi + 1) / len(batches)
        for i in range(iters):
            t0 = time.perf_counter()
            O.iteration(g_sd, d_sd, dec_sd, og, od, batch, 2 + i, cfg)
            times.append(time.perf_counter() - t0)
            if len(times) >= 2 and time.perf_counter() - t_start > share:
                break
        legs[B] = (statistics.median(times), len(times))
    B_main = batches[-1]
    return {"value": round(B_main / legs[B_main][0], 4), "unit": "samples/s", "cores": n_threads, "kind": "port",
            "cpu_model": _cpu_model(),
            "by_batch": {str(B): {"samples_per_s": round(B / dt, 4), "s_per_iteration": round(dt, 3), "iterations": n}
                         for B, (dt, n) in legs.items()},
            "sample": f"oracle.codec_oracle.iteration (torch CPU, fp32) on the same step at T={T} (S={T*320}): B=" +
                      " and B=".join(f"{B} (median of {n} iterations after 1 warm-up, {dt:.2f} s each)" for B, (dt, n) in legs.items())
                      + f"; {n_threads} threads; value = B={B_main}"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the newest committed PMC summary (profiles/*_pmc_traffic.json, written by
    tools/pmc_traffic.py from two rocprofv3 --pmc passes over this same command: FETCH_SIZE x2 and WRITE_SIZE, both KiB, as
    MI355X_MICROARCH.md's HBM section prescribes).  PMC counters cannot be read from inside the process, so the figure is
    the last profiled build's; `traffic_source` names the file."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    if not files:
        return {}
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(kernel_name)
    except (OSError, ValueError, KeyError):
        return {}
    if not k:
        return {}
    out = {"traffic": round(k["hbm_bytes_per_launch"]), "traffic_unit": "bytes/launch (2*FETCH_SIZE + WRITE_SIZE)",
           "traffic_source": "profiles/" + os.path.basename(files[-1])}
    # MFMA utilisation of the same kernels from the newest committed counter pass (tools/pmc_mfma.py:
    # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs))
    mf = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_mfma.json")))
    if mf:
        try:
            with open(mf[-1]) as f:
                kk = json.load(f)["kernels"]
            if kernel_name in kk:
                out["mfma_busy_frac"] = round(kk[kernel_name]["mfma_busy_frac"], 4)
                out["mfma_busy_frac_wgrad_gemm_kernel"] = round(kk.get("wgrad_gemm_kernel", {}).get("mfma_busy_frac", float("nan")), 4)
                out["mfma_busy_source"] = "profiles/" + os.path.basename(mf[-1])
        except (OSError, ValueError, KeyError):
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU: SURVEY.md 8(d) config 2 sweeps {12, 64, 256, max-fit}; 64 is "
                    "the headline, and the reference's default 12 (run_lab3_codec.py:201) is measured too and reported under 'batch12'")
    ap.add_argument("--no-batch12", action="store_true", help="skip the second measurement at the reference's default batch of 12")
    ap.add_argument("--frames", type=int, default=375, help="latent frames T (5 s x 75 Hz); S = 320*T samples")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--stft-stress", action="store_true", help="BASELINE config 5: the six MR-STFT resolutions (256 ... 8192) of SURVEY.md "
                    "8(d); use with --frames 2250 (30 s clips, S = 720000) and a batch sized for the 288 GB of HBM")
    ap.add_argument("--no-fp32-batch12", action="store_true", help="skip the fp32-mode leg (B=12, the rtol-1e-4 parity path) reported under 'fp32_batch12'")
    ap.add_argument("--no-stage2", action="store_true", help="skip the stage-2-like leg (d_step_period=2, R1 every 16 steps, gamma=1: SURVEY.md "
                    "8(d) config 2 reports it separately from the period-1 / no-R1 headline) reported under 'stage2_like'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--cpu-budget-s", type=float, default=75.0)
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel code path (RCCL group, bucketed all-reduce) even with one rank")
    ap.add_argument("--conv-table", action="store_true", help="print per-shape convolution timings of one step to stderr")
    ap.add_argument("--dp-overlap", action="store_true", help="all-reduce bucket by bucket during the backward pass (opt-in: not yet run on more than one GPU)")
    ap.add_argument("--no-dp-overlap", action="store_true", help="all-reduce after the backward pass (the default)")
    args = ap.parse_args()
    # the contract is ONE line on stdout: libraries that print banners there (RCCL prints its version at init) go to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dp = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        if world == 1 and "MASTER_ADDR" not in os.environ:  # --force-dp outside torchrun: a one-rank RCCL group
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("nccl", device_id=device)
        from gan_amd.dp import DataParallel
        dp = DataParallel()
        if args.dp_overlap:
            dp.overlap = True
        elif args.no_dp_overlap:
            dp.overlap = False  # (DataParallel's own default)

    from gan_amd import ops
    from gan_amd.lab3_codec_models import CodecTrainWeights
    from gan_amd.lab3_codec_train import CodecStageTrainConfig, codec_train_iteration
    from gan_amd.optim import FusedAdamW

    G, D, codec = make_models(args.precision, device, seed=328)
    scfg = CodecStageTrainConfig(stage_name="stage2", epochs=1, d_step_period=1, r1_gamma=0.0, g_grad_clip_norm=5.0,
                                 d_grad_clip_norm=5.0, weights=CodecTrainWeights(content=0.0, style=0.0))
    if args.stft_stress:
        scfg.wave_mrstft_resolutions = ((256, 64, 256), (512, 128, 512), (1024, 256, 1024), (2048, 512, 2048),
                                        (4096, 1024, 4096), (8192, 2048, 8192))
    opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    if dp is not None:
        dp.broadcast_params(opt_g, opt_d)
    gen = torch.Generator().manual_seed(328 + rank)

    state = {"batches": None, "i": 0, "models": (G, D, codec, opt_g, opt_d), "cfg": scfg}

    def step(_i=None):
        i = state["i"]
        state["i"] += 1
        batches = state["batches"]
        b = batches[i % len(batches)]
        G_, D_, codec_, og_, od_ = state["models"]
        return codec_train_iteration(state["cfg"], G_, D_, codec_, og_, od_, b[0], b[1], b[2], b[3], b[4], b[5], global_step=i + 1, dp=dp)

    def barrier():
        if dp is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def lstm_flag_on_any_rank() -> bool:
        """The sticky error word of the persistent LSTM launches (read and cleared), OR-ed over the ranks so that all of them take
        the same branch."""
        f = 1.0 if ops.lstm_persist_error(device) else 0.0
        if dp is not None:
            t = torch.tensor([f], device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            f = float(t)
        return f > 0.0

    def measure(batch, steps=None, warmup=None):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
        If a persistent LSTM launch ran out of its bounded wait while gradient all-reduces were overlapped with the backward pass
        (RCCL kernels beside a kernel that needs a workgroup on every CU: never run on hardware before this bench, gan_amd/dp.py),
        every rank drops to the after-backward all-reduce and the measurement is taken again; the line then says so."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        state["batches"] = [make_batch(batch, args.frames, device, gen) for _ in range(2)]
        for attempt in (0, 1):
            for _ in range(warmup):
                out = step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = step()
            barrier()
            el = time.perf_counter() - t0
            if not lstm_flag_on_any_rank():
                break
            if attempt == 0 and dp is not None and dp.overlap:
                dp.overlap = False
                state["overlap_fallback"] = True
                continue
            raise SystemExit("a persistent LSTM launch of the frozen decoder ran out of its bounded wait: the timed steps are invalid")
        if dp is not None:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t)
        lg = float(out["loss_g"])
        assert lg == lg, "loss is NaN"
        return el, lg

    # fp32 mode (the rtol-1e-4 parity path: exact fp32 FMA convolutions) at the reference's default batch, BASELINE.md section 3
    # asks for both precisions; a short leg on its own models, freed before the headline measurement
    fp32_b12 = None
    if args.precision == "bf16" and not args.no_fp32_batch12 and args.frames <= 375 and not args.stft_stress:
        G32, D32, codec32 = make_models("fp32", device, seed=328)
        og32 = FusedAdamW(G32.parameters(), lr=scfg.lr_g, betas=scfg.betas)
        od32 = FusedAdamW(D32.parameters(), lr=scfg.lr_d, betas=scfg.betas)
        if dp is not None:
            dp.broadcast_params(og32, od32)
        state["models"] = (G32, D32, codec32, og32, od32)
        n32 = max(2, min(4, args.steps))
        el32, _ = measure(12, steps=n32, warmup=1)
        fp32_b12 = {"batch_per_gpu": 12, "global_batch": 12 * world, "dtype": "fp32", "value": round(12 * world * n32 / el32, 3),
                    "unit": "samples/s", "ms_per_step": round(1e3 * el32 / n32, 3), "steps": n32, "warmup": 1}
        state["models"] = (G, D, codec, opt_g, opt_d)
        del G32, D32, codec32, og32, od32
        torch.cuda.empty_cache()
    # stage-2-like schedule (run_lab3_codec.py:276-279: D step every second iteration, R1 with gamma 1 every 16th), 16 iterations so
    # that exactly one R1 step is inside the timed region
    stage2 = None
    if not args.no_stage2 and not args.stft_stress:
        import dataclasses
        state["cfg"] = dataclasses.replace(scfg, d_step_period=2, r1_gamma=1.0, r1_interval=16)
        state["i"] = 0
        el2, _ = measure(args.batch, steps=16, warmup=2)
        stage2 = {"d_step_period": 2, "r1_interval": 16, "r1_gamma": 1.0, "batch_per_gpu": args.batch, "steps": 16, "warmup": 2,
                  "value": round(args.batch * world * 16 / el2, 3), "unit": "samples/s", "ms_per_step": round(1e3 * el2 / 16, 3)}
        state["cfg"] = scfg
        state["i"] = 0
    batch12 = None
    if args.batch != 12 and not args.no_batch12:  # the reference's default batch size, second line of the report
        el12, _ = measure(12)
        batch12 = {"batch_per_gpu": 12, "global_batch": 12 * world, "value": round(12 * world * args.steps / el12, 3),
                   "unit": "samples/s", "ms_per_step": round(1e3 * el12 / args.steps, 3)}
    # SURVEY.md 8(d) config 2 sweeps the batch {12, 64, 256, max-fit}: one more point, twice the headline batch (throughput saturates
    # there: profiles/r03d_batch_sweep.txt), reported under 'batch128' and never mixed into `value`
    batch128 = None
    if args.batch == 64 and not args.no_batch12 and not args.stft_stress and args.frames <= 375:
        n128 = max(2, min(4, args.steps))
        el128, _ = measure(128, steps=n128, warmup=1)
        batch128 = {"batch_per_gpu": 128, "global_batch": 128 * world, "value": round(128 * world * n128 / el128, 3), "unit": "samples/s",
                    "ms_per_step": round(1e3 * el128 / n128, 3), "steps": n128, "warmup": 1}
        torch.cuda.empty_cache()
    elapsed, loss_g = measure(args.batch)

    # per-kernel accounting: one more step with HIP events (torch current stream == launch stream) around every
    # convolution launch; achieved = launched multiply-add flops / event time, per kernel.
    roofline = None
    kernels = {}
    # (every rank takes this step -- it contains the gradient all-reduces -- and rank 0 keeps the events)
    if rank == 0:
        ops.PROFILE = []
    if rank == 0 or world > 1:
        step()
        torch.cuda.synchronize()
    if rank == 0:
        prof, ops.PROFILE = ops.PROFILE, None
        if args.conv_table:
            rows = {}
            for kid, _launched, e0, e1, shape, flops in prof:  # algorithmic flops
                r = rows.setdefault((kid,) + shape, [0, 0.0, 0.0])
                r[0] += 1; r[1] += flops; r[2] += e0.elapsed_time(e1)
            print("kernel  (B, Cin, M, J, Lout, nphase, in_stride, pre_act)  launches  ms  TFLOP/s", file=sys.stderr)
            for key, r in sorted(rows.items(), key=lambda kv: -kv[1][2]):
                print(f"{kernel_name(key[0]):38s} {str(key[1:]):44s} {r[0]:3d} {r[2]:8.3f} {r[1] / max(r[2], 1e-9) / 1e9:8.1f}", file=sys.stderr)
        esz = 2 if args.precision == "bf16" else 4
        for kid, launched, e0, e1, shape, flops in prof:
            k = kernels.setdefault(kid, {"launches": 0, "flops": 0.0, "launched": 0.0, "ms": 0.0, "bytes": 0.0})
            k["launches"] += 1
            k["flops"] += flops
            k["launched"] += launched
            k["ms"] += e0.elapsed_time(e1)
            b_, cin_, m_, j_, lout_, _nph, istr_ = shape[:7]
            # algorithmic bytes of one launch: the input window once, the packed weights once, the output once
            k["bytes"] += esz * (b_ * lout_ * max(1, istr_) * cin_ + m_ * j_ * cin_ + b_ * lout_ * m_)
        dom = max((k for k in kernels if k in (3, 4, 7, 8, 9, 11, 12) or k >= 0x1000), key=lambda k: kernels[k]["ms"], default=None)
        if dom is not None:
            k = kernels[dom]
            ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": kernel_name(dom), "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                        "launches_per_step": k["launches"], "avg_launch_us": round(1e3 * k["ms"] / k["launches"], 2),
                        "flops_per_launch_avg": k["flops"] / k["launches"],
                        "launched_TFLOP/s": round(k["launched"] / (k["ms"] * 1e-3) / 1e12, 2),
                        "flops": "algorithmic (conv taps k, DFT K once); launched_TFLOP/s counts the padded taps of strided "
                                 "data gradients and the 3x-deep K of the bf16 hi/lo MR-STFT GEMM",
                        "algorithmic_bytes_per_launch_avg": k["bytes"] / k["launches"]}
            roofline.update(pmc_traffic(kernel_name(dom)))
            # the whole gathered-GEMM family (every conv_gemm_kernel instantiation + the ping-pong kernel) next to its dominant member
            fam = [v for k, v in kernels.items() if k >= 0x1000 or k in (7, 8, 9, 11)]
            fam_ms, fam_fl = sum(v["ms"] for v in fam), sum(v["flops"] for v in fam)
            # D blocks 2-4 alone (k41 stride-4 forward convs, their 11-tap folded data gradients), whichever kernel ran them: the
            # layers the north star's ">= 50 % of peak on the G/D conv kernels" is about
            dl = [(fl, e0.elapsed_time(e1)) for kid, _l, e0, e1, sh, fl in prof
                  if (kid >= 0x1000 or kid == 11) and ((sh[3] == 41 and sh[6] == 4 and sh[1] >= 64) or (sh[3] == 11 and sh[6] == 1 and sh[1] >= 128))]
            if dl:
                d_fl, d_ms = sum(a for a, _ in dl), sum(b for _, b in dl)
                roofline["d_blocks_2_4_fwd_dgrad"] = {"launches": len(dl), "ms": round(d_ms, 3), "TFLOP/s": round(d_fl / d_ms / 1e9, 2),
                                                      "frac": round(d_fl / d_ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
            roofline["gathered_gemm_family"] = {"launches": sum(v["launches"] for v in fam), "ms": round(fam_ms, 3),
                                                "TFLOP/s": round(fam_fl / max(fam_ms, 1e-9) / 1e9, 2),
                                                "frac": round(fam_fl / max(fam_ms, 1e-9) / 1e9 / PEAK_BF16_TFLOPS, 4)}

    if rank == 0:
        total_samples = args.batch * world * args.steps
        line = {
            "metric": "G+D train samples/sec", "value": round(total_samples / elapsed, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "lab3 codec GAN full-size codec shape (T=%d latent frames, S=%d samples), G 256ch x10 "
                                   "blocks, 3-scale D, frozen SEANet decoder; d_step_period=1, no R1, losses adv+FM+latent+"
                                   "continuity+MR-STFT%s, clip 5 + AdamW" % (args.frames, args.frames * 320,
                                                                             " (six resolutions 256..8192)" if args.stft_stress else ""),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "random_init_weights": True},
            "roofline": roofline,
            "kernel_breakdown": {kernel_name(k): {"launches": v["launches"], "ms": round(v["ms"], 3),
                                                    "TFLOP/s": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)}
                                 for k, v in sorted(kernels.items())},
            "loss_g_last": round(loss_g, 5),
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
            "reserved_hbm_gb": round(torch.cuda.max_memory_reserved(device) / 1e9, 2),
            "alloc_retries": int(torch.cuda.memory_stats(device).get("num_alloc_retries", 0)),
        }
        if batch12 is not None:
            line["batch12"] = batch12
        if batch128 is not None:
            line["batch128"] = batch128
        if fp32_b12 is not None:
            line["fp32_batch12"] = fp32_b12
        if stage2 is not None:
            line["stage2_like"] = stage2
        if dp is not None:
            line["config"]["dp_allreduce"] = ("overlapped with the backward pass" if dp.overlap else
                                              "after the backward pass" + (" (fell back: a persistent LSTM launch lost its residency beside the "
                                                                           "overlapped collectives)" if state.get("overlap_fallback") else ""))
        line["config"]["note"] = ("headline batch is 64 per GPU since round 2 (round 1's BENCH line was B=12): compare rounds on "
                                  "'batch12', the reference's default batch, which every run reports")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.frames, (4, 12), args.cpu_iters, args.cpu_budget_s)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dp is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
