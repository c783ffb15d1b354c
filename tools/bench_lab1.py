import sys, torch
sys.path.insert(0, ".")
from gan_amd.lab1_adapter import Lab1WaveAdapter
dev = torch.device("cuda")
ad = Lab1WaveAdapter(24000, 22050, 256, dev)
for B in (12, 64):
    x = (0.1 * torch.randn(B, 1, 120000, device=dev)).requires_grad_(True)
    def f():
        m = ad(x); m.mean().backward(); x.grad = None
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B}: adapter fwd+bwd {e0.elapsed_time(e1)/10:.3f} ms")
