import sys, torch, ctypes as C
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = torch.randn(64, 120000, 32, device="cuda").bfloat16(); b = torch.randn_like(a)
out = torch.zeros(1, device="cuda")
t = timeit(lambda: L.check(L.lib().ga_reduce(a.data_ptr(), b.data_ptr(), L.dt(a), a.numel(), L.RED_ABSDIFF, 1.0 / a.numel(), out.data_ptr(), L.stream()), "r"))
out.zero_(); L.lib().ga_reduce(a.data_ptr(), b.data_ptr(), L.dt(a), a.numel(), L.RED_ABSDIFF, 1.0 / a.numel(), out.data_ptr(), L.stream())
print(f"absdiff 2x{a.numel()*2/1e6:.0f} MB: {t*1e3:.1f} us {2*a.numel()*2/t/1e9:.2f} TB/s  value {out.item():.6f} ref {(a.float()-b.float()).abs().mean().item():.6f}")
g = torch.randn(34_000_000, device="cuda")
t = timeit(lambda: L.lib().ga_grad_sqnorm(g.data_ptr(), g.numel(), out.data_ptr(), L.stream()))
out.zero_(); L.lib().ga_grad_sqnorm(g.data_ptr(), g.numel(), out.data_ptr(), L.stream())
print(f"sqnorm {g.numel()*4/1e6:.0f} MB: {t*1e3:.1f} us {g.numel()*4/t/1e9:.2f} TB/s value {out.item():.1f} ref {g.double().square().sum().item():.1f}")
