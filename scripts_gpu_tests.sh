#!/bin/bash
# run the GPU test files one after another; stop as soon as one ends by a signal / timeout (not a plain test failure)
mkdir -p gpurun_out
for f in "$@"; do
  name=$(basename "$f" .py)
  timeout -k 10 420 python -u -m pytest "$f" -m gpu -v --timeout 180 --timeout-method thread -x --tb=short -rP > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "$f -> exit $rc"; tail -n 25 "gpurun_out/$name.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: abnormal exit $rc"; exit $rc; fi
done
