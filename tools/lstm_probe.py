"""fp32 and bf16 LstmSkip forward + input gradient on fixed inputs -> gpurun_out/lstm_probe_<tag>.pt (compare two builds bit by bit)"""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import ops
tag = sys.argv[1]
torch.manual_seed(3)
lstm = torch.nn.LSTM(512, 512, 2).cuda()
out = {}
for dtype in (torch.float32, torch.bfloat16):
    for persist in (True, False):
        ops.LSTM_PERSIST = persist
        w = ops.prepare_lstm_weights(lstm, dtype)
        for B, T, ng in ((2, 16, 2), (5, 9, 3), (40, 7, 40)):
            g = torch.Generator(device="cuda").manual_seed(B * 100 + T)
            x = (torch.randn(B, T, 512, device="cuda", generator=g) * 0.7).to(dtype).requires_grad_(True)
            y = ops.LstmSkip.apply(x, w, ng)
            gy = torch.randn(B, T, 512, device="cuda", generator=g).to(dtype)
            gy[ng:] = 0
            gx, = torch.autograd.grad(y, x, gy)
            out[f"{dtype}_{persist}_{B}_{T}"] = (y.detach().float().cpu(), gx.float().cpu())
torch.save(out, f"gpurun_out/lstm_probe_{tag}.pt")
if len(sys.argv) > 2:
    ref = torch.load(f"tools/lstm_probe_{sys.argv[2]}.pt")
    for k in out:
        print(k, "y equal", torch.equal(out[k][0], ref[k][0]), "gx equal", torch.equal(out[k][1], ref[k][1]),
              "max dgx", (out[k][1] - ref[k][1]).abs().max().item())
