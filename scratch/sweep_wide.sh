#!/bin/bash
# GPU box: correctness with the 256x128 tile forced for every M%256==0 shape, then bench per mode / threshold
set -e
for w in 2 3; do
  GA_GEMM_WIDE=$w GA_GEMM_WIDE_MIN=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -x -q --timeout 120 --timeout-method thread > gpurun_out/wide_tests_$w.log 2>&1
  tail -1 gpurun_out/wide_tests_$w.log
done
for cfg in "0 600" "2 600" "3 600" "2 300" "3 300" "2 150"; do
  set -- $cfg
  GA_GEMM_WIDE=$1 GA_GEMM_WIDE_MIN=$2 timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_wide_$1_$2.json 2> gpurun_out/bench_wide_$1_$2.err
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_wide_$1_$2.json")); print("wide=$1 min=$2", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_breakdown"].items() if "gemm" in k}, flush=True)
PY
done
