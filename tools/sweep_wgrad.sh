#!/bin/bash
set -e
for m in 768 512 1024 1536 2048; do
  GA_WGRAD_TARGET=$m timeout -k 10 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_wt_$m.json 2> gpurun_out/bench_wt_$m.err
  python - <<PY
import json; d=json.load(open("gpurun_out/bench_wt_$m.json")); print("wgrad_target=$m", d["value"], d["ms_per_step"], flush=True)
PY
done
