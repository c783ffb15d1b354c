import os, sys, torch
sys.path.insert(0, ".")
from gan_amd import ops
lstm = torch.nn.LSTM(512, 512, 2).cuda()
w = ops.prepare_lstm_weights(lstm, torch.bfloat16)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in (16, 64, 128, 192, 256):
    x = torch.randn(B, 375, 512, device="cuda").bfloat16()
    with torch.no_grad():
        tn = t(lambda: ops.LstmSkip.apply(x, w))
    xg = x.clone().requires_grad_(True)
    ta = t(lambda: ops.LstmSkip.apply(xg, w, B))
    print(f"B={B}: fwd no-grad {tn:.3f} ms ({tn/376*1e3:.2f} us/step)   saving all gates {ta:.3f} ms ({ta/376*1e3:.2f} us/step)", flush=True)
# the step's shape: 256 items, the first 64 need gradients; and its backward
x = torch.randn(256, 375, 512, device="cuda").bfloat16().requires_grad_(True)
tm = t(lambda: ops.LstmSkip.apply(x, w, 64))
y = ops.LstmSkip.apply(x, w, 64)
gy = torch.randn_like(y)
def fb():
    y = ops.LstmSkip.apply(x, w, 64)
    torch.autograd.grad(y, x, gy)
tfb = t(fb)
print(f"B=256 / 64 saved: fwd {tm:.3f} ms ({tm/376*1e3:.2f} us/step)   fwd+bwd {tfb:.3f} ms (bwd {(tfb-tm)/376*1e3:.2f} us/step)", flush=True)
