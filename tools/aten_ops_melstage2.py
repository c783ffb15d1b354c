"""Host-side view of the mel-track stage-2 iteration (bf16, reference sizes; the same set-up as tools/bench_melstage2.py): the second epoch of a
train_stage2 run under the torch profiler -- operator counts, CPU time and device time per iteration.  usage: aten_ops_melstage2.py [batch] [all]"""
import sys, tempfile, time
from pathlib import Path
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from gan_amd import lab3_models as PM, lab3_train as PT
from gan_amd.lab3_bridge import ChunkEncoder, FrozenLab1Encoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ALL = len(sys.argv) > 2 and sys.argv[2] == "all"
dev = torch.device("cuda")
torch.manual_seed(0)
n_genres, n_batches = 4, 12
tmp = Path(tempfile.mkdtemp())
enc_net = ChunkEncoder(n_sources=n_genres)
torch.save({"cfg": {"sample_rate": 22050, "chunk_seconds": 5.0, "z_dim": 128}, "source_to_idx": {f"g{i}": i for i in range(n_genres)},
            "model": enc_net.state_dict()}, str(tmp / "lab1.pt"))
enc = FrozenLab1Encoder(tmp / "lab1.pt", device="cuda", precision="bf16")
G = PM.ReconstructionDecoder(precision="bf16").to(dev)
D = PM.MelDiscriminator(cond_dim=160, precision="bf16").to(dev)
w = PT.TrainWeights(mrstft=0.5)
if ALL:
    w = PT.TrainWeights(adv=0.5, recon_l1=8.0, content=2.0, style=1.0, continuity=1.0, mrstft=0.5, flatness=0.5, feature_match=2.0, perceptual=0.7,
                        style_hinge=0.6, contrastive=0.4, batch_infonce_diversity=0.3, pfm_style=1.5, diversity=0.8, timbre_balance=0.9,
                        lowmid_recon=0.6, spectral_tilt=0.7, zcr_proxy=1.1, hf_muzzle=1.3, highpass_anchor=0.45, style_mid=0.55, mel_diversity=0.65,
                        target_profile=3.0)
cfg = PT.StageTrainConfig(stage_name="stage2", epochs=2, weights=w)
batches = []
for _ in range(n_batches):
    gi = torch.randint(0, n_genres, (B,))
    batches.append({"mel_norm": torch.rand(B, 96, 256) * 2 - 1, "z_content": F.normalize(torch.randn(B, 128), dim=-1),
                    "z_style": F.normalize(torch.randn(B, 128), dim=-1), "genre_idx": gi})
batches = [{k: v.to(dev) for k, v in b.items()} for b in batches]


class Loader:
    dataset = None

    def __iter__(self):
        return iter(batches)

    def __len__(self):
        return len(batches)


cond_bank = F.normalize(torch.randn(n_genres, 160), dim=-1)
protos = torch.randn(n_genres, 128)
exemplars = {g: F.normalize(torch.randn(16, 128), dim=-1) for g in range(n_genres)}
profile = torch.rand(n_genres, 96) + 0.05
profile = profile / profile.sum(dim=1, keepdim=True)
times = []
orig_save = PT._save_stage_checkpoint


def timed_save(*a, **k):  # one call per epoch: marks the epoch boundary (the file write itself is left out of the timing)
    torch.cuda.synchronize()
    times.append(time.perf_counter())


from torch.profiler import profile as tprofile, ProfilerActivity
prof = tprofile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])


def timed_save(*a, **k):  # epoch boundary: start profiling after the first epoch
    torch.cuda.synchronize()
    times.append(time.perf_counter())
    if len(times) == 1:
        prof.__enter__()
    else:
        prof.__exit__(None, None, None)


PT._save_stage_checkpoint = timed_save
hist = PT.train_stage2(G, D, enc, Loader(), cond_bank, protos, exemplars, profile, dev, cfg, tmp / "s2.pt")
ev = prof.key_averages()
rows = sorted(((e.count / n_batches, e.self_cpu_time_total / n_batches, e.self_device_time_total / n_batches, e.key) for e in ev), reverse=True)
print(f"per iteration: {sum(r[0] for r in rows if r[3].startswith('aten::')):.0f} aten calls, "
      f"{sum(r[1] for r in rows) / 1e3:.1f} ms self CPU time in profiled ops, {sum(r[2] for r in rows) / 1e3:.1f} ms device time")
for c, cpu, dv, k in rows[:28]:
    print(f"{c:7.1f} calls  {cpu:8.1f} us cpu  {dv:8.1f} us device  {k[:70]}")
print("--- by CPU time")
for c, cpu, dv, k in sorted(rows, key=lambda r: -r[1])[:14]:
    print(f"{c:7.1f} calls  {cpu:8.1f} us cpu  {dv:8.1f} us device  {k[:70]}")
per_it = (times[1] - times[0]) / n_batches
print(f"stage 2{' (all 23 terms)' if ALL else ''}: B={B} {per_it * 1e3:.2f} ms per iteration = {B / per_it:.0f} samples/s "
      f"(second epoch of {n_batches} iterations; loss_g {hist[-1]['loss_g']:.4f}, loss_d {hist[-1]['loss_d']:.4f})")
