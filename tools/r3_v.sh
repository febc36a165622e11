#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q -m gpu -k "camera_heights or fuzz or extreme or non_default or other_camera" > gpurun_out/r3v_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3v_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 260 python tools/fuzz_parity.py 300 9101 flat 2>&1 | tail -2
tools/kprof.sh hcam tools/hcam_bench.py 36,29127 32,32768 28,37449 24,43690 27,38836 40,26214 | grep -v cast
