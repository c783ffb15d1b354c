"""Is the step host-bound or GPU-bound?  Host issue time of K steps (no synchronisation inside) against their wall time to completion."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from gan_amd.lab3_codec_train import CodecStageTrainConfig, codec_train_iteration
from gan_amd.lab3_codec_models import CodecTrainWeights
from gan_amd.optim import FusedAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
G, D, codec = bench.make_models("bf16", dev, 328)
cfg = CodecStageTrainConfig(stage_name="stage3", epochs=1, d_step_period=1, r1_gamma=0.0, weights=CodecTrainWeights(content=0.0, style=0.0))
og, od = FusedAdamW(G.parameters(), lr=cfg.lr_g, betas=cfg.betas), FusedAdamW(D.parameters(), lr=cfg.lr_d, betas=cfg.betas)
gen = torch.Generator().manual_seed(1)
b = bench.make_batch(B, 375, dev, gen)
def step(i):
    return codec_train_iteration(cfg, G, D, codec, og, od, b[0], b[1], b[2], b[3], b[4], b[5], global_step=i + 1)
for i in range(4): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host issue {1e3 * (t1 - t0) / K:.2f} ms/step, wall {1e3 * (t2 - t0) / K:.2f} ms/step, GPU still busy {1e3 * (t2 - t1):.2f} ms after the last issue")
