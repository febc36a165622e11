#!/bin/bash
# round 3, GPU call I: whole parity suite, then unit vs flat store kernels on the unit kernels' geometries
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3i_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3i_pytest.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|Mismatch|FAILED" gpurun_out/r3i_pytest.log | head -20; exit 1; fi
export TOPSHAPES_STEPS=60
: > gpurun_out/r3i.txt
for flat in 0 1; do
  export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so RCW_TOP_FLAT=$flat
  for shape in 12,12,32,256 8,8,16,256 10,10,32,256 9,9,32,256 16,16,32,256; do
    tools/kprof.sh "flat${flat}_$shape" tools/top_view_shapes.py $shape >> gpurun_out/r3i.txt 2>&1 || exit 1
  done
done
grep -E "store" gpurun_out/r3i.txt
