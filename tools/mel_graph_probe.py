"""The mel-track stage-1 iteration replayed from a recorded HIP graph (gan_amd.lab3_train.GraphedStage1) against the eager iteration at
the reference's default sizes, bf16: time per iteration by batch, and the loss after the same number of iterations on the same batch.
usage: mel_graph_probe.py [batch ...]"""
import copy, sys, time, torch
sys.path.insert(0, ".")
from gan_amd import lab3_models as PM, lab3_train as PT
from gan_amd.optim import FusedAdamW
dev = torch.device("cuda")
N = 26
for B in [int(a) for a in sys.argv[1:]] or [32, 64]:
    torch.manual_seed(0)
    G0 = PM.ReconstructionDecoder(precision="bf16").to(dev).train()
    D0 = PM.MelDiscriminator(cond_dim=160, precision="bf16").to(dev).train()
    cfg = PT.StageTrainConfig(stage_name="stage1", epochs=1, weights=PT.TrainWeights(content=0.0, style=0.0))
    real = torch.rand(B, 96, 256, device=dev) * 2 - 1
    zc, cond = torch.randn(B, 128, device=dev), torch.randn(B, 160, device=dev)
    res = {}
    for mode in ("eager", "graph"):
        G, D = copy.deepcopy(G0), copy.deepcopy(D0)
        og = FusedAdamW(G.parameters(), lr=cfg.lr_g, betas=cfg.betas, weight_decay=0.0)
        od = FusedAdamW(D.parameters(), lr=cfg.lr_d, betas=cfg.betas, weight_decay=0.0)
        run = PT.GraphedStage1(cfg, G, D, og, od) if mode == "graph" else (lambda r, z, c, i: PT.mel_stage1_iteration(cfg, G, D, og, od, r, z, c, i))
        for i in range(6):
            out = run(real, zc, cond, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(6, N):
            out = run(real, zc, cond, i)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / (N - 6), float(out["loss_g"]), float(out["loss_d"]))
    (te, ge, de), (tg, gg, dg) = res["eager"], res["graph"]
    print(f"B={B}: eager {te * 1e3:.2f} ms ({B / te:.0f} samples/s)   graph replay {tg * 1e3:.2f} ms ({B / tg:.0f} samples/s)   "
          f"after {N} iterations loss_g {ge:.4f} / {gg:.4f}, loss_d {de:.4f} / {dg:.4f}", flush=True)
