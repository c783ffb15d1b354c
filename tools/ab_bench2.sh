#!/bin/bash
# usage: tools/ab_bench2.sh "VAR=A VAR2=B" "VAR=C" ... [-- rounds]   interleaved bench runs of several environments (fresh process each)
n=${AB_ROUNDS:-2}
envs=()
for a in "$@"; do envs+=("$a"); done
for i in $(seq $n); do
  for e in "${envs[@]}"; do
    env $e timeout -k 10 200 python bench.py --no-batch12 --no-fp32-batch12 --no-stage2 --no-cpu-baseline --steps 10 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('$e', d['value'], d['ms_per_step'])"
  done
done
