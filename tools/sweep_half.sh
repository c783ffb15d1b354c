#!/bin/bash
# usage: tools/sweep_half.sh  (GPU box) - correctness with the 64x128 tiles forced wide, then bench per threshold
set -e
GA_GEMM_HALF_MAX=100000 timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -x -q --timeout 120 --timeout-method thread > gpurun_out/half_tests.log 2>&1
for h in 0 100 180 300 600; do
  GA_GEMM_HALF_MAX=$h timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_h$h.json 2> gpurun_out/bench_h$h.err
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_h$h.json")); print("half_max=$h", d["value"], d["ms_per_step"], flush=True)
PY
done
