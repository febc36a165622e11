#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TOPSHAPES_STEPS=40
: > gpurun_out/r3e.txt
for shape in 32,32,32,1024 24,24,32,256 16,16,32,256; do for runs in 1 2 3 4; do
  echo "runs=$runs" >> gpurun_out/r3e.txt
  TOPSHAPES_RUNS=$runs timeout -k 10 120 python tools/top_view_shapes.py $shape 2>&1 | grep map >> gpurun_out/r3e.txt
done; done
cat gpurun_out/r3e.txt
