#!/bin/bash
# Round 5, GPU box: rcw_fill_flat_kernel with two wavefronts to a slot of the window (development library, RCW_FILL_FLAT_PAIRS=1)
# against the shipped form: parity (the camera-height tests), then kernel times by rocprofv3 over camera heights.
R=$PWD; export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out; : > gpurun_out/r05_flat_pairs.txt
RCW_FILL_FLAT_PAIRS=1 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "odd_camera_heights or beyond_2_32 or flat_kernel_geometries" 2>&1 | tail -3 | tee -a gpurun_out/r05_flat_pairs.txt
for pairs in 0 1; do
  RCW_FILL_FLAT_PAIRS=$pairs HCAM_STEPS=120 tools/kprof.sh "flat_pairs$pairs" tools/hcam_bench.py 40,26214 84,12483 100,10486 250,4194 300,3495 480,2184 36,29127 2>&1 | grep -E "fill_flat|failed" | sed "s/^/pairs=$pairs /" >> gpurun_out/r05_flat_pairs.txt
  grep -h "^H_cam" gpurun_out/kp_flat_pairs$pairs.log | sed "s/^/pairs=$pairs /" >> gpurun_out/r05_flat_pairs.txt
done
cat gpurun_out/r05_flat_pairs.txt
