#!/bin/bash
# usage: tools/ab_bench_b.sh <batch> "VAR=A" "VAR=B" ...   interleaved bench runs at one batch size (AB_ROUNDS rounds, fresh process each)
b=$1; shift
n=${AB_ROUNDS:-2}
for i in $(seq $n); do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --batch $b --no-batch12 --no-fp32-batch12 --no-stage2 --no-cpu-baseline --steps 20 --warmup 4 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('$e', d['value'], d['ms_per_step'])"
  done
done
