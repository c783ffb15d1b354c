import os, sys, torch
sys.path.insert(0, ".")
from gan_amd import ops
lstm = torch.nn.LSTM(512, 512, 2).cuda()
w = ops.prepare_lstm_weights(lstm, torch.bfloat16)
for B, ng in ((36, 12), (256, 64), (128, 32)):
    x = torch.randn(B, 375, 512, device="cuda").bfloat16().requires_grad_(True)
    for _ in range(2): y = ops.LstmSkip.apply(x, w, ng)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = ops.LstmSkip.apply(x, w, ng)
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 5
    g = torch.randn_like(y)
    for _ in range(2): torch.autograd.grad(y, x, g, retain_graph=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5): torch.autograd.grad(y, x, g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / 5
    print(f"affine={os.environ.get('GA_LSTM_AFFINE')} B={B} grad {ng}: fwd {tf:.3f} ms ({tf/376*1e3:.2f} us/step)  bwd {tb:.3f} ms ({tb/376*1e3:.2f} us/step) err={ops.lstm_persist_error()}", flush=True)
