#!/bin/bash
# usage: tools/ab_bench.sh VAR A B [rounds]   interleaved bench runs with VAR=A and VAR=B (same box, fresh process each)
var=$1; a=$2; b=$3; n=${4:-3}
for i in $(seq $n); do
  for v in $a $b; do
    env $var=$v timeout -k 10 200 python bench.py --no-batch12 --no-cpu-baseline --steps 10 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('$var=$v', d['value'], d['ms_per_step'])"
  done
done
