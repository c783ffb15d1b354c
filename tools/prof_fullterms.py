"""One codec step with EVERY generator term on (style judge CE + Lab1 content / style terms through the waveform -> log-mel adapter),
bench shape, for a kernel trace: which kernels the f1 terms add.  usage: prof_fullterms.py [batch] [steps]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from gan_amd import lab3_codec_train as T
from gan_amd.lab3_codec_judge import CodecStyleJudge
from gan_amd.lab3_codec_models import CodecTrainWeights
from gan_amd.lab1_adapter import Lab1WaveAdapter
from gan_amd.optim import FusedAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda")


class Lab1:  # a frozen encoder of the ChunkEncoder topology (lab3_bridge.py:34-76): torch's own Conv2d, or this build's on the HIP kernels
    class cfg:
        sample_rate = 22050

    def __init__(self, hip: bool):
        torch.manual_seed(5)
        from gan_amd.lab3_bridge import ChunkEncoder
        enc = ChunkEncoder(n_sources=5, z_dim=128, precision="bf16")
        if not hip:  # the same parameters in the reference's plain-torch topology
            import types
            nn = torch.nn
            ref = nn.Module()
            ref.backbone = nn.Sequential(nn.Conv2d(1, 32, 3, 2, 1), nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 64, 3, 2, 1), nn.BatchNorm2d(64), nn.ReLU(),
                                         nn.Conv2d(64, 128, 3, 2, 1), nn.BatchNorm2d(128), nn.ReLU(), nn.AdaptiveAvgPool2d((1, 1)))
            ref.shared, ref.content_head, ref.style_head = nn.Linear(128, 256), nn.Linear(256, 128), nn.Linear(256, 128)
            ref.load_state_dict({k: v for k, v in enc.state_dict().items() if k.split(".")[0] in ("backbone", "shared", "content_head", "style_head")})
            F = torch.nn.functional

            def fwd(self, m, grl_lambda=0.0):
                h = F.relu(self.shared(self.backbone(m.unsqueeze(1)).flatten(1)))
                return {"shared_feat": h, "z_content": F.normalize(self.content_head(h), dim=-1), "z_style": F.normalize(self.style_head(h), dim=-1)}
            ref.forward = types.MethodType(fwd, ref)
            enc = ref
        self.model = enc.to(dev).eval()
        for p in self.model.parameters():
            p.requires_grad = False

    def forward_log_mel_tensor(self, m):
        return self.model(m, grl_lambda=0.0)


G, D, codec = bench.make_models("bf16", dev, seed=328)
judge = CodecStyleJudge(128, 4, precision="bf16").to(dev).eval()
adapter = Lab1WaveAdapter(24000, 22050, 256, dev)
import os
MODES = ("none", "codec_judge_ce+lab1 (torch encoder)", "codec_judge_ce+lab1 (HIP encoder)")
for mode in ([MODES[int(os.environ["FULLTERMS_MODE"])]] if "FULLTERMS_MODE" in os.environ else MODES):
    w = CodecTrainWeights(content=0.0, style=0.0) if mode == "none" else CodecTrainWeights()
    scfg = T.CodecStageTrainConfig(stage_name="stage2", epochs=1, d_step_period=1, g_grad_clip_norm=5.0, d_grad_clip_norm=5.0, weights=w,
                                   style_loss_mode="codec_judge_ce")
    og = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    od = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    b = bench.make_batch(B, 375, dev, torch.Generator().manual_seed(2))
    tgt = torch.randint(0, 4, (B,), device=dev)
    src = torch.randint(0, 4, (B,), device=dev)
    lab1 = None if mode == "none" else Lab1("HIP" in mode)
    kw = {} if mode == "none" else dict(style_judge=judge, tgt_genre_idx=tgt, src_genre_idx=src, lab1_encoder=lab1, lab1_adapter=adapter,
                                        zs_src=torch.nn.functional.normalize(torch.randn(B, 128, device=dev), dim=-1))
    for i in range(3):
        out = T.codec_train_iteration(scfg, G, D, codec, og, od, *b, global_step=i + 1, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = T.codec_train_iteration(scfg, G, D, codec, og, od, *b, global_step=i + 4, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{mode}: B={B} {dt * 1e3:.2f} ms per iteration; " + ", ".join(f"{k}={float(v):.4f}" for k, v in out.items() if k in ("loss_g", "loss_content", "loss_style")))
