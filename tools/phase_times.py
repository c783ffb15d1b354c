"""Event-timed phases of one training iteration (bench configuration)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from gan_amd import lab3_codec_train as T
from gan_amd.lab3_codec_models import CodecTrainWeights
from gan_amd.optim import FusedAdamW
dev = torch.device("cuda")
G, D, codec = bench.make_models("bf16", dev, seed=328)
scfg = T.CodecStageTrainConfig(stage_name="stage2", epochs=1, d_step_period=1, r1_gamma=0.0, g_grad_clip_norm=5.0,
                               d_grad_clip_norm=5.0, weights=CodecTrainWeights(content=0.0, style=0.0))
opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
gen = torch.Generator().manual_seed(328)
b = bench.make_batch(12, 375, dev, gen)
rec = []
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        rec.append((label, e0, e1)); return r
    setattr(obj, name, w)
wrap(G, "forward", "G.forward")
wrap(codec, "decode_embeddings_mixed", "decoder fwd (12 grad + 24 frozen)")
wrap(D, "forward", "D.forward")
wrap(torch.Tensor, "backward", "backward")
wrap(opt_g, "step", "opt_g.step"); wrap(opt_d, "step", "opt_d.step")
wrap(opt_g, "gather_grads", "opt_g.gather"); wrap(opt_d, "gather_grads", "opt_d.gather")
wrap(T, "_waveform_mrstft_loss", "mrstft fwd")
wrap(T, "multiscale_feature_matching_loss", "fm fwd")
wrap(T, "latent_losses", "latent fwd")
for i in range(4):
    rec.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    T.codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, b[0], b[1], b[2], b[3], b[4], b[5], global_step=i + 1)
    e1.record(); torch.cuda.synchronize()
tot = e0.elapsed_time(e1)
acc = 0.0
for label, a, c in rec:
    t = a.elapsed_time(c); acc += t
    print(f"{label:40s} {t:7.3f} ms")
print(f"{'sum of phases':40s} {acc:7.3f} ms   iteration {tot:7.3f} ms")
