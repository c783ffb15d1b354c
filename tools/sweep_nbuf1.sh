#!/bin/bash
set -e
for m in 600 300 450 800 1100; do
  GA_GEMM_NBUF1_MIN=$m timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_n1_$m.json 2> gpurun_out/bench_n1_$m.err
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_n1_$m.json")); print("nbuf1_min=$m", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_breakdown"].items() if "gemm" in k}, flush=True)
PY
done
