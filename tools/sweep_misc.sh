#!/bin/bash
set -e
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_ms_$name.json 2> gpurun_out/bench_ms_$name.err
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_ms_$name.json")); print("$name", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_breakdown"].items() if "gemm" in k}, flush=True)
PY
}
run base GA_X=0
run deep200 GA_GEMM_DEEP_MAX=200
run deep400 GA_GEMM_DEEP_MAX=400
run half64 GA_GEMM_HALF_MAX=64
run half220 GA_GEMM_HALF_MAX=220
run mg4 GA_GEMM_MGROUP=4
run mg16 GA_GEMM_MGROUP=16
