"""List the small aten ops of one iteration (fills, copies, adds, cats) by input shape, to find avoidable autograd glue."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gan_amd import lab3_codec_train as T
from gan_amd.lab3_codec_models import CodecTrainWeights
from gan_amd.optim import FusedAdamW
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
G, D, codec = bench.make_models("bf16", dev, seed=1)
scfg = T.CodecStageTrainConfig(stage_name="stage2", epochs=1, d_step_period=1, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
                               weights=CodecTrainWeights(content=0.0, style=0.0))
opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
b = bench.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 12, 375, dev, torch.Generator().manual_seed(2))
for i in range(3):
    T.codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, *b, global_step=i + 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    T.codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, *b, global_step=4)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        where = [f for f in e.stack if "gan_amd" in f or "bench.py" in f]
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:70], " <- ".join(w.split("/")[-1].strip() for w in where[:2])))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {tot:.0f} us over {sum(r[1] for r in rows)} calls")
for t, c, k, sh, w in rows[:70]:
    print(f"{t:8.1f} us x{c:<3d} {k:22s} {sh:70s} {w}")
