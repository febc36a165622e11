#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TOPSHAPES_STEPS=60
: > gpurun_out/r3e.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forms_write and two" > gpurun_out/r3e_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/r3e_pytest.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|Mismatch" gpurun_out/r3e_pytest.log | head -20; exit 1; fi
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  for shape in 8,8,24,256 8,8,13,256 8,8,10,256 9,9,32,256 8,8,12,256; do
    tools/kprof.sh "$(basename $v .so)_$shape" tools/top_view_shapes.py $shape >> gpurun_out/r3e.txt 2>&1 || exit 1
  done
done
grep -E "store" gpurun_out/r3e.txt
