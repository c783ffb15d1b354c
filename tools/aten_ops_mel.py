"""The aten ops (glue around the HIP kernels) of one bf16 mel-track stage-1 iteration, by device time.  usage: aten_ops_mel.py [batch]"""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import lab3_models as PM, lab3_train as PT
from gan_amd.optim import FusedAdamW
from torch.profiler import profile, ProfilerActivity
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
torch.manual_seed(0)
G = PM.ReconstructionDecoder(precision="bf16").to(dev).train()
D = PM.MelDiscriminator(cond_dim=160, precision="bf16").to(dev).train()
cfg = PT.StageTrainConfig(stage_name="stage1", epochs=1, weights=PT.TrainWeights(content=0.0, style=0.0))
og = FusedAdamW(G.parameters(), lr=cfg.lr_g, betas=cfg.betas, weight_decay=0.0)
od = FusedAdamW(D.parameters(), lr=cfg.lr_d, betas=cfg.betas, weight_decay=0.0)
real = torch.rand(B, 96, 256, device=dev) * 2 - 1
zc, cond = torch.randn(B, 128, device=dev), torch.randn(B, 160, device=dev)
for i in range(3):
    PT.mel_stage1_iteration(cfg, G, D, og, od, real, zc, cond, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    PT.mel_stage1_iteration(cfg, G, D, og, od, real, zc, cond, 3)
    torch.cuda.synchronize()
rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:80]) for e in prof.key_averages(group_by_input_shape=True)
        if e.self_device_time_total > 0]
rows.sort(reverse=True)
aten = [r for r in rows if r[2].startswith("aten::")]
print(f"device time: all ops {sum(r[0] for r in rows):.0f} us, aten ops {sum(r[0] for r in aten):.0f} us over {sum(r[1] for r in aten)} calls")
by = {}
for t, c, k, _ in aten:
    by[k] = (by.get(k, (0, 0))[0] + t, by.get(k, (0, 0))[1] + c)
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{t:9.1f} us x{c:<4d} {k}")
print()
for t, c, k, sh in aten[:40]:
    print(f"{t:8.1f} us x{c:<3d} {k:24s} {sh}")
