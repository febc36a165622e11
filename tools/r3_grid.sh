#!/bin/bash
# round 3: the flat kernels with more than one workgroup per CU (development library's RCW_TOP_STORE_GRID / RCW_FILL_GRID)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/r3_grid.txt
export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so TOPSHAPES_STEPS=40
for g in 256 384 512 768 1024; do
  for shape in 8,8,20,256 8,8,13,256 8,8,10,256 9,9,32,256; do
    RCW_TOP_STORE_GRID=$g timeout -k 10 200 tools/kprof.sh "grid$g-$shape" tools/top_view_shapes.py $shape 2>&1 | grep "store" >> gpurun_out/r3_grid.txt
  done
  RCW_FILL_GRID=$g timeout -k 10 200 tools/kprof.sh "fillgrid$g" tools/hcam_bench.py 100,10486 40,26214 300,3495 250,4194 2>&1 | grep "flat" >> gpurun_out/r3_grid.txt
done
cat gpurun_out/r3_grid.txt
