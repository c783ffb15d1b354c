#!/bin/bash
# usage: tools/timeline_run.sh <tag> ["ENV=V ..."] ["extra bench args"]   one rocprofv3 kernel trace of 6 bench steps -> gpurun_out/<tag>_timeline.txt (on the GPU box)
tag=$1; envs=$2; extra=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/tl_$tag
rm -rf $out; mkdir -p $out
for e in $envs; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/kt -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch12 --no-fp32-batch12 --no-stage2 $extra > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
python tools/prof_timeline.py $(find $out/kt -name "*.db" | head -1) gpurun_out/${tag}_timeline.txt
rm -rf $out
