import sys; sys.path.insert(0,'.')
import torch, numpy as np
from oracle import codec_oracle as O
from oracle.procfill import proc_input
from tests.gpu_common import build_gpu_models
g_sd,d_sd,dec_sd = O.fresh_state()
_, D, _ = build_gpu_models(g_sd,d_sd,dec_sd,"fp32")
wav = proc_input((2,1,16384),"donly.wav",0,0.8)
# oracle
dp={k:v.clone() for k,v in d_sd.items()}
wr = wav.clone().requires_grad_(True)
outs = O.discriminator_forward(dp, wr)
tot = sum(lg.mean() + sum(f.square().mean() for f in fs) for lg,fs in outs)
ref, = torch.autograd.grad(tot, wr)
wg = wav.cuda().requires_grad_(True)
outs_g = D(wg)
for s,((lg,fs),(lr,fr)) in enumerate(zip(outs_g,outs)):
    for j,(a,b) in enumerate(zip(fs,fr)):
        e=(a.detach().cpu()-b.detach()).abs()
        print("scale",s,"feat",j,"max err",e.max().item(), "rel", (e.max()/b.abs().max()).item(), "argmax", np.unravel_index(e.argmax().item(), e.shape))
tg = sum(lg.float().mean() + sum(f.float().square().mean() for f in fs) for lg,fs in outs_g)
tg.backward()
err=(wg.grad.cpu()-ref).abs()[ :,0]
bad=(err>2e-4*ref.abs().max()).nonzero()
print("nbad",len(bad),"of",err.numel(),"max",err.max().item()); print(bad[:40].tolist()); print(bad[-10:].tolist())
