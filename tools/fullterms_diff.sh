#!/bin/bash
# kernel-trace the full-terms codec step in two modes and list what the second adds (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for m in 0 2; do
  out=gpurun_out/ft_$m; rm -rf $out; mkdir -p $out
  FULLTERMS_MODE=$m timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o p -- python3 tools/prof_fullterms.py 64 12 > $out/kt.log 2>&1 || { tail -3 $out/kt.log; exit 1; }
  python tools/prof_summary.py $(find $out/kt -name "*.db" | head -1) gpurun_out/ft_stats_$m.csv 15 > /dev/null
  rm -rf $out
done
