"""prints g_grad_norm / d_grad_norm of iteration 4 of the fp32 golden sequence (tests/test_gpu_step.py) a few times"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_step as T
from test_gpu_step import O, build_gpu_models, run_gpu_iteration, make_inputs
golden = np.load("tests/golden/codec_golden.npz")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    g_sd, d_sd, dec_sd = O.fresh_state()
    G, D, codec = build_gpu_models(g_sd, d_sd, dec_sd, "fp32")
    cfg = O.StepCfg(d_step_period=2, r1_gamma=1.0, r1_interval=2,
                    weights=O.Weights(adv=0.55, latent_l1=0.2, latent_continuity=1.0, content=0.0, style=0.0, mrstft=0.0, feature_match=1.0, delta_budget=1.0))
    og = od = None
    row = []
    for it in range(1, 5):
        out = run_gpu_iteration(G, D, codec, make_inputs(2, 16, f"seq{it}"), cfg, it, og, od)
        og, od = out["opt_g"], out["opt_d"]
        row.append(f"it{it} g {float(out['g_grad_norm']):.5f}/{float(golden[f'seq/it{it}/g_grad_norm']):.5f}")
    print(" | ".join(row), flush=True)
