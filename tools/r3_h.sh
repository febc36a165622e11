#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "camera_heights or extreme or non_default" > gpurun_out/r3h_pytest.log 2>&1; rc=$?; tail -2 gpurun_out/r3h_pytest.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|Mismatch" gpurun_out/r3h_pytest.log | head -20; exit 1; fi
: > gpurun_out/r3h.txt
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  tools/kprof.sh "$(basename $v .so)" tools/hcam_bench.py 100,10486 84,12483 250,4194 300,3495 40,26214 >> gpurun_out/r3h.txt 2>&1
done
grep -E "fill_flat" gpurun_out/r3h.txt
