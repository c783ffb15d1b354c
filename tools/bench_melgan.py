"""Mel-track stage-1 iteration time (reference defaults: decoder base 256, discriminator base 64 with the projection head, 96 x 256
mels, bce loss, adv + L1 + MR-STFT terms) on the HIP-backed modules, fp32 and bf16.  usage: bench_melgan.py [batch] [fp32|bf16]"""
import sys, time, torch
sys.path.insert(0, ".")
from gan_amd import lab3_models as PM, lab3_train as PT
from gan_amd.optim import FusedAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
for prec in ((sys.argv[2],) if len(sys.argv) > 2 else ("fp32", "bf16")):
    torch.manual_seed(0)
    G = PM.ReconstructionDecoder(precision=prec).to(dev).train()
    D = PM.MelDiscriminator(cond_dim=160, precision=prec).to(dev).train()
    cfg = PT.StageTrainConfig(stage_name="stage1", epochs=1, weights=PT.TrainWeights(content=0.0, style=0.0))
    og = FusedAdamW(G.parameters(), lr=cfg.lr_g, betas=cfg.betas, weight_decay=0.0)
    od = FusedAdamW(D.parameters(), lr=cfg.lr_d, betas=cfg.betas, weight_decay=0.0)
    real = torch.rand(B, 96, 256, device=dev) * 2 - 1
    zc, cond = torch.randn(B, 128, device=dev), torch.randn(B, 160, device=dev)
    for i in range(3):
        PT.mel_stage1_iteration(cfg, G, D, og, od, real, zc, cond, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        out = PT.mel_stage1_iteration(cfg, G, D, og, od, real, zc, cond, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{prec}: B={B} {dt * 1e3:.2f} ms per iteration = {B / dt:.0f} samples/s (loss_g {float(out['loss_g']):.4f}, loss_d {float(out['loss_d']):.4f})")
