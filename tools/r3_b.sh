#!/bin/bash
# round 3, GPU call B: attribution of the flat top store kernel's chunk-loop cost (development build, RCW_TOP_DEBUG bits)
set -o pipefail
mkdir -p gpurun_out
export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so
: > gpurun_out/r3b.txt
for shape in 8,8,24,256 8,8,13,256 9,9,32,256; do
  for dbg in 0 16 32 48 64 112 128; do
    echo "== shape $shape RCW_TOP_DEBUG=$dbg" >> gpurun_out/r3b.txt
    RCW_TOP_DEBUG=$dbg timeout -k 10 120 python tools/top_view_shapes.py $shape >> gpurun_out/r3b.txt 2>&1 || exit 1
  done
done
grep -v amdgpu.ids gpurun_out/r3b.txt
