#!/bin/bash
# round 3: the balanced draw kernel — parity, then the draw kernel's time with and without it (development library's RCW_TOP_DRAW_BALANCED=0)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "top" > gpurun_out/r3draw_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3draw_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for mode in top flat; do timeout -k 10 200 python tools/fuzz_parity.py 100 4$RANDOM $mode 2>&1 | tail -1; done
export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so TOPSHAPES_STEPS=40
for shape in 8,8,32,256 32,32,32,1024 16,16,32,256 8,16,32,512 24,24,32,256 8,8,13,256; do
  for bal in 1 0; do
    RCW_TOP_DRAW_BALANCED=$bal tools/kprof.sh "bal$bal-$shape" tools/top_view_shapes.py $shape 2>&1 | grep "draw"
    grep -h "^map" gpurun_out/kp_bal$bal-$shape.log | cut -c1-150
  done
done
