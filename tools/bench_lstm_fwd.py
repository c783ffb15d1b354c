import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_amd import ops
lstm = torch.nn.LSTM(512, 512, 2).cuda()
w = ops.prepare_lstm_weights(lstm, torch.bfloat16)
for B in (36, 12):
    x = torch.randn(B, 375, 512, device="cuda").bfloat16()
    with torch.no_grad():
        for _ in range(2): ops.LstmSkip.apply(x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.LstmSkip.apply(x, w)
        e1.record(); torch.cuda.synchronize()
    print(f"B={B}: fwd {e0.elapsed_time(e1)/5:.3f} ms  ({e0.elapsed_time(e1)/5/376*1e3:.2f} us/step)")
# with saved gates (training step: 12 gradient items of 36)
x = torch.randn(36, 375, 512, device="cuda").bfloat16().requires_grad_(True)
for ng in (36, 12):
    for _ in range(2): ops.LstmSkip.apply(x, w, ng)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.LstmSkip.apply(x, w, ng)
    e1.record(); torch.cuda.synchronize()
    print(f"B=36 saving gates for {ng}: fwd {e0.elapsed_time(e1)/5:.3f} ms  ({e0.elapsed_time(e1)/5/376*1e3:.2f} us/step)")
