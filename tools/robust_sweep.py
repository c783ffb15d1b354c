"""Shape sweep of the full iteration (bf16): finite losses for odd batch sizes / frame counts."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gan_amd import lab3_codec_train as T
from gan_amd.lab3_codec_models import CodecTrainWeights
from gan_amd.optim import FusedAdamW
dev = torch.device("cuda")
for prec in ("bf16", "fp32"):
    G, D, codec = bench.make_models(prec, dev, seed=1)
    scfg = T.CodecStageTrainConfig(stage_name="stage2", epochs=1, d_step_period=1, r1_gamma=1.0, r1_interval=2, g_grad_clip_norm=5.0,
                                   d_grad_clip_norm=5.0, weights=CodecTrainWeights(content=0.0, style=0.0))
    opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    gen = torch.Generator().manual_seed(2)
    shapes = ((1, 375), (5, 101), (12, 375), (3, 37), (17, 200)) if prec == "bf16" else ((2, 101), (3, 37))
    for B, frames in shapes:
        b = bench.make_batch(B, frames, dev, gen)
        for i in range(2):
            out = T.codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, b[0], b[1], b[2], b[3], b[4], b[5], global_step=i + 1)
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in out.items() if v.dim() == 0}
        ok = all(v == v and abs(v) < 1e6 for v in vals.values())
        print(prec, B, frames, "OK" if ok else "BAD", {k: round(v, 4) for k, v in vals.items() if k in ("loss_g", "loss_d", "loss_mrstft", "r1", "g_grad_norm", "d_grad_norm")}, flush=True)
        assert ok
print("sweep ok")
