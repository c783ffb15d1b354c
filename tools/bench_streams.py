"""Do two latency-bound LSTM chains on two HIP streams overlap?"""
import sys, time, torch
sys.path.insert(0, ".")
from gan_amd.lab3_codec_bridge import FrozenEncodec
codec = FrozenEncodec(random_init=True, device="cuda", precision="bf16")
dec = codec.decoder
qa = 0.5 * torch.randn(12, 128, 375, device="cuda")
qb = 0.5 * torch.randn(36, 128, 375, device="cuda")
with torch.no_grad():
    dec(qa); dec(qb)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def a():
    with torch.no_grad(): dec(qa)
def b():
    with torch.no_grad(): dec(qb)
def both():
    with torch.cuda.stream(s1):
        with torch.no_grad(): dec(qa)
    with torch.cuda.stream(s2):
        with torch.no_grad(): dec(qb)
def both_seq():
    a(); b()
print(f"decode B=12: {run(a):.2f} ms   B=36: {run(b):.2f} ms   sequential: {run(both_seq):.2f} ms   two streams: {run(both):.2f} ms")
# host-only launch cost: how long does the host take to enqueue one decode?
torch.cuda.synchronize(); t0 = time.perf_counter(); a(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host enqueue time of one B=12 decode: {(t1 - t0) * 1e3:.2f} ms")
