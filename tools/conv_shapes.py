"""A few D-layer conv launches, five of each, for a per-shape rocprofv3 --pmc FETCH_SIZE pass (tools/pmc_shapes.sh)."""
import sys, torch
sys.path.insert(0, ".")
from gan_amd import _lib as L, ops
from gan_amd.ops import ConvGeom
B = 128
torch.manual_seed(0)
for name, g, Ln in (("b2", ConvGeom("conv", 64, 128, 41, 4, 20, 1), 30000), ("b3", ConvGeom("conv", 128, 256, 41, 4, 20, 1), 7500),
                    ("b4", ConvGeom("conv", 256, 512, 41, 4, 20, 1), 1875)):
    x = torch.randn(B, Ln, g.cin, device="cuda").bfloat16()
    w = torch.randn(*g.wdims, device="cuda") / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device="cuda")
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device="cuda").bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    for _ in range(5):
        ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)
    for _ in range(5):
        ops.conv_dgrad(g, gy, wd, Ln, dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=x)
    for _ in range(5):
        ops.conv_wgrad(g, gy, x)
    torch.cuda.synchronize()
    print(name, "fwd in MB", x.numel() * 2 / 1e6, "out MB", gy.numel() * 2 / 1e6, flush=True)
