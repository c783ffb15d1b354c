#!/bin/bash
# usage: tools/sweep_env.sh BATCH "ENV1" "ENV2" ...   one bench run per environment (interleaved twice), batch per GPU given
b=$1; shift
for i in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 200 python bench.py --batch $b --no-batch12 --no-fp32-batch12 --no-stage2 --no-cpu-baseline --steps 10 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('B=$b', '$e', d['value'], d['ms_per_step'])"
  done
done
