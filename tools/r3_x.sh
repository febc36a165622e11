#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3x_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3x_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for w in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 10 2>/dev/null > gpurun_out/r3x_$w.json; python tools/bench_brief.py $w < gpurun_out/r3x_$w.json
done
