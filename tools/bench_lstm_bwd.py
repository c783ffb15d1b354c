import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_amd import ops
lstm = torch.nn.LSTM(512, 512, 2).cuda()
w = ops.prepare_lstm_weights(lstm, torch.bfloat16)
for B in (12,):
    x = torch.randn(B, 375, 512, device="cuda").bfloat16().requires_grad_(True)
    y = ops.LstmSkip.apply(x, w)
    g = torch.randn_like(y)
    for _ in range(2): torch.autograd.grad(y, x, g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): torch.autograd.grad(y, x, g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print(f"UPB={os.environ.get('GA_LSTM_BUPB')} B={B}: bwd {e0.elapsed_time(e1)/5:.3f} ms  ({e0.elapsed_time(e1)/5/376*1e3:.2f} us/step)")
