"""Probe (run by hand on the GPU box from the repo root): run-to-run margin of the ps_batch decoder parity check of tests/test_gpu_melgan.py.
Lives under tests/ because it imports the test module (and with it the oracle's procedural fill)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_melgan as M
tag = "ps_batch"
kw = M.DEC_CASES[tag]
worst = []
for r in range(12):
    G = M.load_filled(M.PM.ReconstructionDecoder(**kw), 11)
    zc = M.proc_input((2, kw.get("zc_dim", 128)), f"melgan.zc.{tag}", 0).to(M.DEV)
    v = M.proc_input((2, kw.get("cond_dim", 160)), f"melgan.v.{tag}", 0).to(M.DEV)
    mel = G(zc, v)
    got = mel[:, ::5, ::17].detach().cpu().numpy().astype(np.float64)
    want = np.asarray(M.GOLD[f"dec/{tag}/out_slice"], dtype=np.float64)
    err = np.abs(got - want)
    worst.append(float((err - 1e-4 * np.abs(want)).max()))
print("max over the slice of err - 1e-4|want| in 12 runs:", " ".join(f"{w:.2e}" for w in worst))
