#!/bin/bash
# usage: tools/stalls_run.sh <steps> ["ENV=V ..."]   (on the GPU box)
steps=$1; envs=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/st_run
rm -rf $out; mkdir -p $out
for e in $envs; do export "$e"; done
timeout -k 10 400 rocprofv3 --kernel-trace -d $out/kt -o p -- python3 bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-batch12 --no-fp32-batch12 --no-stage2 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
tail -1 $out/kt.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench under trace', d['value'], d['ms_per_step'])"
python tools/prof_stalls.py $(find $out/kt -name "*.db" | head -1)
rm -rf $out
