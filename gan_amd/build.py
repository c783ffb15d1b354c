"""Build libgan_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libgan_amd.so"
SOURCES = ["capi.hip", "conv_direct.hip", "conv_mfma.hip", "conv_gemm.hip", "conv_pp.hip", "conv_win.hip", "conv_stream.hip", "conv_swin.hip", "resunit.hip", "conv_coop.hip", "wgrad_mfma.hip", "wgrad_gemm.hip", "wgrad_cin1_mfma.hip", "thin.hip", "lstm.hip", "lstm_persist.hip", "elementwise.hip", "cnorm.hip", "stft.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [CSRC / "common.h", CSRC / "swin_common.h", HERE.parent / "include" / "gan_amd.h"]
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = objdir / (src + ".o")
        if force or _stale(obj, [CSRC / src] + headers):
            jobs.append([hipcc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [str(objdir / (s + ".o")) for s in SOURCES]
    if force or jobs or not LIB.exists():
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
