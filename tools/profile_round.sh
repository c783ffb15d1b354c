#!/bin/bash
# usage: tools/profile_round.sh <tag> [bench args...]   (on the GPU box; writes summaries to gpurun_out/<tag>_*)
# One rocprofv3 run per pass: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the MFMA-busy counters, each alone.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
args="--steps 6 --warmup 2 --no-cpu-baseline --no-batch12 --no-fp32-batch12 --no-stage2 $*"
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o p -- python3 bench.py $args > $out/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $out/kt.log; exit 1; }
db=$(find $out/kt -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${tag}_kernel_stats.csv 9 | tee gpurun_out/${tag}_kernel_summary.txt
tail -1 $out/kt.log > gpurun_out/${tag}_bench_line_under_trace.json
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout -s KILL 400 rocprofv3 --pmc $ctrs --output-format csv -d $out/$name -o p -- python3 bench.py $args > $out/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $out/$name.log; exit 1; }
  echo "pass $name done: $(find $out/$name -name '*counter_collection.csv' | head -1)"
done
f=$(find $out/fetch -name '*counter_collection.csv' | head -1); w=$(find $out/write -name '*counter_collection.csv' | head -1); m=$(find $out/mfma -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $f $w gpurun_out/${tag}_pmc_traffic.json
python tools/pmc_mfma.py $m gpurun_out/${tag}_pmc_mfma.json
rm -rf $out
