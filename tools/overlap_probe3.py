"""Trace target: a 192-item frozen decode on a side stream while generator-sized convs (25 KiB LDS budget) are issued on the main
stream; tools/prof_corun.py lists what ran while the persistent LSTM kernel was resident."""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
from gan_amd.lab3_codec_bridge import FrozenEncodec
dev = "cuda"
torch.manual_seed(0)
codec = FrozenEncodec(random_init=True, device=dev, precision="bf16")
q = 0.5 * torch.randn(192, 128, 375, device=dev)
g = ConvGeom("conv", 256, 256, 3, 1, 1, 1)
x = torch.randn(64, 375, 256, device=dev).bfloat16()
wf = ops.pack_weight(g, torch.randn(*g.wdims, device=dev) / 27.7, g.fwd_pack, torch.bfloat16)
bias = torch.randn(256, device=dev)
side = torch.cuda.Stream(device=dev)
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
for rep in range(3):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side), torch.no_grad():
        y = codec.decode_embeddings(q)
    old = L.lib().ga_set_gemm_lds_budget(budget)
    for _ in range(150):
        ops.conv_forward(g, x, wf, bias)
    L.lib().ga_set_gemm_lds_budget(old)
    main.wait_stream(side)
    torch.cuda.synchronize()
