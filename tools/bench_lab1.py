"""Forward + backward time of the Lab1 waveform -> log-mel adapter: HIP path vs the torch-op form.  usage: bench_lab1.py B"""
import sys, torch
sys.path.insert(0, ".")
from gan_amd.lab1_adapter import Lab1WaveAdapter, wave_to_logmel_torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda")
ad = Lab1WaveAdapter(24000, 22050, 256, dev)
x = (0.1 * torch.randn(B, 1, 120000, device=dev)).requires_grad_(True)
def run(fn):
    m = fn(x); g, = torch.autograd.grad(m.sum(), x); return g
for name, fn in (("hip", ad), ("torch", lambda w: wave_to_logmel_torch(w, 24000, 22050, 256))):
    run(fn); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(fn)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/10:.3f} ms fwd+bwd at B={B}")
