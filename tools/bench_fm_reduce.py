"""Feature-matching reductions in isolation: ga_reduce(|fake - real|) on the five feature maps of one discriminator scale (B items each
half of a [real ; fake] tensor), operands rotated through several copies so that nothing is cache-warm.  usage: bench_fm_reduce.py [B] [scale]"""
import sys
import torch
sys.path.insert(0, ".")
from gan_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
S = 120000 >> sc
shapes = [(S, 32), (S // 4, 64), (S // 16, 128), (S // 64, 256), ((S + 255) // 256, 512)]
dev = torch.device("cuda")
lib = L.lib()
out = torch.zeros(1, device=dev)
tot_b = tot_t = 0.0
for rows, ch in shapes:
    n = B * rows * ch
    ncopy = max(2, int(1.5e9 // (4 * n)) + 1)
    bufs = [torch.randn(2 * n, device=dev).to(torch.bfloat16) for _ in range(ncopy)]
    def run(k):
        t = bufs[k % ncopy]
        L.check(lib.ga_reduce(t[n:].data_ptr(), t.data_ptr(), L.DT_BF16, n, L.RED_ABSDIFF, 1.0 / n, out.data_ptr(), L.stream()), "ga_reduce")
    for k in range(3):
        run(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for k in range(reps):
        run(k)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"rows {rows:7d} ch {ch:4d}: {4 * n / 1e6:8.1f} MB  {us:8.1f} us  {4 * n / us / 1e6:6.2f} TB/s")
    tot_b += 4 * n
    tot_t += us
print(f"scale {sc}: {tot_b / 1e6:.0f} MB in {tot_t:.0f} us = {tot_b / tot_t / 1e6:.2f} TB/s")
