"""What runs BESIDE the persistent LSTM launch of a 192-item frozen decode?  Main-stream candidates: tiny elementwise kernels (no LDS),
generator-sized gathered GEMMs with the usual tiles (48 KiB of LDS) and under a 25 KiB LDS budget (24 KiB tiles)."""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
from gan_amd.lab3_codec_bridge import FrozenEncodec
items = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dev = "cuda"
torch.manual_seed(0)
codec = FrozenEncodec(random_init=True, device=dev, precision="bf16")
q = 0.5 * torch.randn(items, 128, 375, device=dev)
g = ConvGeom("conv", 256, 256, 3, 1, 1, 1)
x = torch.randn(64, 375, 256, device=dev).bfloat16()
w = torch.randn(*g.wdims, device=dev) / (g.cin * g.k) ** 0.5
wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
bias = torch.randn(256, device=dev)
small = torch.randn(64, 375, 256, device=dev)
def ew(n=300):
    for _ in range(n): small.mul_(1.0001)
def convs(n=60):
    for _ in range(n): ops.conv_forward(g, x, wf, bias)
def convs_budget(n=60):
    old = L.lib().ga_set_gemm_lds_budget(25600)
    try:
        convs(n)
    finally:
        L.lib().ga_set_gemm_lds_budget(old)
def decode():
    with torch.no_grad():
        return codec.decode_embeddings(q)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
a = timed(decode)
print(f"decode({items}) alone {a:.2f} ms", flush=True)
side = torch.cuda.Stream(device=dev)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for name, fn in (("elementwise", lambda: ew(300 * N)), ("G convs, usual tiles", lambda: convs(60 * N)), ("G convs, 25 KiB budget", lambda: convs_budget(60 * N))):
    b = timed(fn)
    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            y = decode()
        fn()
        main.wait_stream(side)
        return y
    c = timed(both)
    print(f"{name:30s}: alone {b:.2f} ms, beside the decode {c:.2f} ms -> {a + b - c:+.2f} ms hidden", flush=True)
print("lstm error flag:", ops.lstm_persist_error(dev))
