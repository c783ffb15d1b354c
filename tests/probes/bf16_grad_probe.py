"""Where does the bf16 G gradient differ from the fp32 oracle?  Per loss-weight subset, and the oracle's own sensitivity."""
import sys, torch
sys.path.insert(0, ".")
from oracle import codec_oracle as O
from tests.gpu_common import build_gpu_models, stage_cfg_from_oracle
from tests.helpers import make_inputs
from gan_amd.lab3_codec_train import codec_train_iteration
from gan_amd.optim import FusedAdamW

def flat_ref(want, key, mod, opt):
    ref = torch.zeros(opt.numel, dtype=torch.float64)
    names = [k for k, p in mod.named_parameters() if p.requires_grad]
    for k, p, off in zip(names, opt.params, opt.offsets):
        ref[off:off + p.numel()] = want[key][k].double().reshape(-1)
    return ref

def run(weights, precision, B=2, T=32, perturb=None):
    g_sd, d_sd, dec_sd = O.fresh_state()
    batch = list(make_inputs(B, T, "bf16"))
    cfg = O.StepCfg(weights=weights)
    ob = [t.clone() for t in batch]
    if perturb:
        ob[0] = ob[0] * (1 + perturb * torch.randn_like(ob[0]))
    want = O.iteration({k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}, dec_sd,
                       O.AdamWState(), O.AdamWState(), ob, 1, cfg, record=True)
    if precision is None:
        return want
    G, D, codec = build_gpu_models(g_sd, d_sd, dec_sd, precision)
    scfg = stage_cfg_from_oracle(cfg)
    opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    grabbed = {}
    for name, opt in (("g", opt_g), ("d", opt_d)):
        orig = opt.step
        def step(grads_gathered=False, _o=orig, _opt=opt, _n=name):
            grabbed[_n] = _opt.flat_g.detach().clone()
            return _o(grads_gathered=grads_gathered)
        opt.step = step
    dev = [t.cuda() for t in batch]
    codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, *dev, 1)
    out = {}
    for name, mod, opt, key in (("d", D, opt_d, "d_grads"), ("g", G, opt_g, "g_grads")):
        flat = grabbed[name].double().cpu()
        ref = flat_ref(want, key, mod, opt)
        out[name] = (float(torch.dot(flat, ref) / (flat.norm() * ref.norm())), float((flat - ref).norm() / ref.norm()), float(ref.norm()))
    return out

W = O.Weights
torch.manual_seed(0)
for label, w in (("all (default)", W(content=0.0, style=0.0)),
                 ("no mrstft", W(content=0.0, style=0.0, mrstft=0.0)),
                 ("mrstft only", W(adv=0.0, latent_l1=0.0, latent_continuity=0.0, content=0.0, style=0.0, feature_match=0.0, mrstft=2.0)),
                 ("adv only", W(adv=0.5, latent_l1=0.0, latent_continuity=0.0, content=0.0, style=0.0, feature_match=0.0, mrstft=0.0)),
                 ("fm only", W(adv=0.0, latent_l1=0.0, latent_continuity=0.0, content=0.0, style=0.0, feature_match=1.0, mrstft=0.0)),
                 ("latent only", W(adv=0.0, latent_l1=4.0, latent_continuity=1.0, content=0.0, style=0.0, feature_match=0.0, mrstft=0.0))):
    for prec in ("fp32", "bf16"):
        r = run(w, prec)
        print(f"{label:14s} {prec}: G cos {r['g'][0]:.5f} relL2 {r['g'][1]:.4f} |ref| {r['g'][2]:.4g}   D cos {r['d'][0]:.5f} relL2 {r['d'][1]:.4f}", flush=True)
# the fp32 oracle against itself with q_src perturbed at the bf16 rounding level (2^-9 relative, uniform-ish)
base = run(W(content=0.0, style=0.0), None)
for eps in (1e-6, 1e-4, 2e-3):
    pert = run(W(content=0.0, style=0.0), None, perturb=eps)
    a = torch.cat([v.double().reshape(-1) for v in base["g_grads"].values()])
    b = torch.cat([pert["g_grads"][k].double().reshape(-1) for k in base["g_grads"]])
    print(f"oracle vs oracle, q_src * (1 + {eps} N(0,1)): G cos {float(torch.dot(a, b) / (a.norm() * b.norm())):.5f} relL2 {float((a - b).norm() / a.norm()):.4f}", flush=True)
