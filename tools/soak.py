"""Soak: N full iterations at the bench workload; allocator high-water mark and losses every 50 iterations."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gan_amd import lab3_codec_train as T
from gan_amd.lab3_codec_models import CodecTrainWeights
from gan_amd.optim import FusedAdamW
dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
G, D, codec = bench.make_models("bf16", dev, seed=1)
scfg = T.CodecStageTrainConfig(stage_name="stage2", epochs=1, d_step_period=1, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0,
                               weights=CodecTrainWeights(content=0.0, style=0.0))
opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
gen = torch.Generator().manual_seed(2)
batches = [bench.make_batch(12, 375, dev, gen) for _ in range(4)]
t0 = time.time()
for i in range(N):
    b = batches[i % 4]
    out = T.codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, b[0], b[1], b[2], b[3], b[4], b[5], global_step=i + 1)
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        vals = {k: round(float(out[k]), 4) for k in ("loss_g", "loss_d", "loss_fm", "loss_mrstft")}
        assert all(v == v for v in vals.values())
        print(i + 1, f"{(time.time() - t0) / (i + 1) * 1e3:.2f} ms/it", f"alloc {torch.cuda.memory_allocated() >> 20} MiB",
              f"reserved {torch.cuda.memory_reserved() >> 20} MiB", vals, flush=True)
print("soak ok")
