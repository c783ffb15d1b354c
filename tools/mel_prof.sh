#!/bin/bash
# usage (GPU box): tools/mel_prof.sh <tag>   -- cnorm tests, kernel trace of the bf16 mel-track iteration (B=64), then its untraced time
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && export PYTHONPATH=$PWD || exit 1
timeout -k 10 300 python -u -m pytest tests/test_gpu_cnorm.py -q --timeout 120 --timeout-method thread > gpurun_out/cnorm_test.log 2>&1 || { grep -E "^(FAILED|E  )|passed|failed" gpurun_out/cnorm_test.log | cut -c1-300 | head -20; exit 1; }
tail -1 gpurun_out/cnorm_test.log
out=gpurun_out/prof_mel; rm -rf $out; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/kt -o p -- python3 tools/bench_melgan.py 64 bf16 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
db=$(find $out/kt -name "*.db" | head -1)
python tools/prof_summary.py $db gpurun_out/${tag}_melgan_kernel_stats.csv 13 > gpurun_out/${tag}_melgan_kernel_summary.txt
head -${2:-14} gpurun_out/${tag}_melgan_kernel_summary.txt | cut -c1-150
rm -rf $out
timeout -k 10 200 python -u tools/bench_melgan.py 64 bf16
