"""GPU: does the discriminator's gradient depend on how the batch is cut?  D forward + hinge-like loss + backward on 4 items at once against
the same 4 items as 2 + 2 (gradients summed), per parameter, fp32 and bf16 modes.  usage: python tools/batch_split_check.py [T]"""
import sys
import torch
sys.path.insert(0, ".")
from gan_amd.lab3_codec_models import MultiScaleWaveDiscriminator

dev = torch.device("cuda")
S = 320 * (int(sys.argv[1]) if len(sys.argv) > 1 else 12)
for prec in ("fp32", "bf16"):
    torch.manual_seed(1)
    D = MultiScaleWaveDiscriminator(2, precision=prec).to(dev).train()
    wav = 0.3 * torch.randn(4, 1, S, device=dev)

    def grads(parts):
        for p in D.parameters():
            p.grad = None
        for sl in parts:
            outs = D(wav[sl])
            loss = sum(torch.relu(1.0 - lg.float()).sum() for lg, _ in outs)  # a SUM over elements: additive over any split of the batch
            loss.backward()
        return {n: p.grad.detach().clone().double() for n, p in D.named_parameters()}

    whole, split, again = grads([slice(0, 4)]), grads([slice(0, 2), slice(2, 4)]), grads([slice(0, 4)])
    print(f"--- {prec}, S={S}: per parameter |split - whole| / |whole| (l2), and run-to-run of the whole batch")
    tot_n = tot_d = 0.0
    for n in whole:
        e = float((split[n] - whole[n]).norm() / whole[n].norm().clamp_min(1e-30))
        r = float((again[n] - whole[n]).norm() / whole[n].norm().clamp_min(1e-30))
        tot_n += float((split[n] - whole[n]).norm() ** 2); tot_d += float(whole[n].norm() ** 2)
        if e > 2e-6 or n.endswith("blocks.2.conv.weight_g"):
            print(f"  {n:40s} {tuple(whole[n].shape)!s:18s} split {e:.2e}   rerun {r:.2e}   |g| {float(whole[n].norm()):.3e}")
    print(f"  all parameters: {(tot_n / tot_d) ** 0.5:.2e}")

# the frozen decoder and the generator: outputs of 4 items at once against 2 + 2
from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator
T = S // 320
for prec in ("fp32", "bf16"):
    torch.manual_seed(2)
    codec = FrozenEncodec(random_init=True, device="cuda", precision=prec)
    G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision=prec).to(dev).eval()
    q = 0.5 * torch.randn(4, 128, T, device=dev)
    zc, zs, nz = torch.randn(4, 128, device=dev), torch.randn(4, 128, device=dev), torch.randn(4, 32, device=dev)
    with torch.no_grad():
        xw = codec.decode_embeddings(q).float()
        xs = torch.cat([codec.decode_embeddings(q[:2]), codec.decode_embeddings(q[2:])]).float()
        gw = G(q, zc, zs, noise=nz).float()
        gs = torch.cat([G(q[:2], zc[:2], zs[:2], noise=nz[:2]), G(q[2:], zc[2:], zs[2:], noise=nz[2:])]).float()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"--- {prec}: decode 2+2 vs 4: {rel(xs, xw):.2e} (max {float((xs - xw).abs().max() / xw.abs().max()):.2e});  generator 2+2 vs 4: {rel(gs, gw):.2e}")
    qq = q.clone().requires_grad_(True)
    codec.decode_embeddings(qq).float().square().sum().backward()
    gw_ = qq.grad.clone()
    qa, qb = q[:2].clone().requires_grad_(True), q[2:].clone().requires_grad_(True)
    codec.decode_embeddings(qa).float().square().sum().backward()
    codec.decode_embeddings(qb).float().square().sum().backward()
    print(f"    decoder input gradient 2+2 vs 4: {rel(torch.cat([qa.grad, qb.grad]), gw_):.2e}")
