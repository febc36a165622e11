#!/bin/bash
# round 3, GPU call A: parity suite, then the shape benches of the two flat kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3a_pytest.log 2>&1; rc=$?; tail -15 gpurun_out/r3a_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out"; exit $rc; fi
timeout -k 10 300 python tools/hcam_bench.py > gpurun_out/r3a_hcam.txt 2>&1; rc=$?; cat gpurun_out/r3a_hcam.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "hcam timed out"; exit $rc; fi
timeout -k 10 500 python tools/top_view_shapes.py > gpurun_out/r3a_top_shapes.txt 2>&1; rc=$?; cat gpurun_out/r3a_top_shapes.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "top shapes timed out"; exit $rc; fi
echo done
