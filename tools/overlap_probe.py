"""How well does the frozen decode (latency-bound LSTM + HBM-bound tail) run BESIDE a stream of compute-bound D convolutions?
A: decode alone, B: convs alone, C: decode on a side stream (normal / high priority) while the convs run on the main stream.
usage: overlap_probe.py [items] [conv repeats]"""
import os, sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
from gan_amd.lab3_codec_bridge import FrozenEncodec
items = int(sys.argv[1]) if len(sys.argv) > 1 else 192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda"
torch.manual_seed(0)
codec = FrozenEncodec(random_init=True, device=dev, precision="bf16")
q = 0.5 * torch.randn(items, 128, 375, device=dev)
convs = []
for g, Ln in ((ConvGeom("conv", 64, 128, 41, 4, 20, 1), 30000), (ConvGeom("conv", 128, 256, 41, 4, 20, 1), 7500), (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 1875)):
    x = torch.randn(128, Ln, g.cin, device=dev).bfloat16()
    w = torch.randn(*g.wdims, device=dev) / (g.cin * g.k) ** 0.5
    convs.append((g, x, ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), torch.randn(g.cout, device=dev)))
def run_convs():
    for _ in range(reps):
        for g, x, wf, b in convs:
            ops.conv_forward(g, x, wf, b, post_act=L.ACT_LRELU)
def run_decode():
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
a = timed(run_decode)
b = timed(run_convs)
print(f"decode alone {a:.2f} ms   convs alone {b:.2f} ms   sum {a + b:.2f}", flush=True)
for prio, name in ((0, "normal"), (-1, "high")):
    side = torch.cuda.Stream(device=dev, priority=prio)
    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            y = run_decode()
        run_convs()
        main.wait_stream(side)
        return y
    c = timed(both)
    print(f"side stream ({name} priority): {c:.2f} ms   = {c - max(a, b):+.2f} over the longer one, {a + b - c:.2f} hidden", flush=True)
print("lstm error flag:", ops.lstm_persist_error(dev))
