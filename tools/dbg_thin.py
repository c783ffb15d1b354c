import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from gan_amd import ops, _lib as L
from gan_amd.ops import ConvGeom
torch.manual_seed(0)
g = ConvGeom("conv", 1, 32, 15, 1, 7, 1)
for B, Ln in ((2, 16384), (2, 4096), (3, 5000)):
    w = torch.randn(32, 1, 15)
    gy = torch.randn(B, 32, Ln)
    x = torch.zeros(B, 1, Ln, requires_grad=True)
    y = F.conv1d(x, w, None, padding=7)
    ref, = torch.autograd.grad(y, x, gy)
    wp = ops.pack_weight(g, w.cuda(), g.dgrad_pack, torch.float32)
    dx = ops.conv_dgrad(g, gy.permute(0,2,1).contiguous().cuda(), wp, Ln).cpu().permute(0,2,1)
    err = (dx - ref).abs()
    bad = (err > 1e-4 * ref.abs().max()).nonzero()
    print(B, Ln, "max err", err.max().item(), "nbad", len(bad), bad[:20].tolist())
