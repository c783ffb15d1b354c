import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_amd import _lib as L, ops
from gan_amd.lab3_codec_bridge import FrozenEncodec
codec = FrozenEncodec(random_init=True, device=torch.device("cuda"), precision="bf16")
lstm = [m for m in codec.decoder.modules() if type(m).__name__.lower().find("lstm") >= 0]
print([type(m).__name__ for m in lstm])
q = torch.randn(36, 128, 375, device="cuda")
with torch.no_grad():
    for _ in range(3):
        y = codec.decode_embeddings(q)
torch.cuda.synchronize()
buf = (C.c_longlong * 1024)()
lib = L.lib()
lib.ga_dbg_read.argtypes = [C.c_void_p]
print("rc", lib.ga_dbg_read(buf))
t = torch.tensor(list(buf), dtype=torch.float64).view(256, 4)
t0 = t[:, 0].min()
rel = (t - t0) * 10.0 / 1000.0
for name, sl in (("layer0", slice(0, 128)), ("layer1", slice(128, 256))):
    r = rel[sl]
    print(name, f"start {r[:,0].min():.2f}..{r[:,0].max():.2f}  loads+mma mean {(r[:,1]-r[:,0]).mean():.2f} max {(r[:,1]-r[:,0]).max():.2f}  "
          f"reduce {(r[:,2]-r[:,1]).mean():.2f}  epilogue mean {(r[:,3]-r[:,2]).mean():.2f} max {(r[:,3]-r[:,2]).max():.2f}  end max {r[:,3].max():.2f}")
