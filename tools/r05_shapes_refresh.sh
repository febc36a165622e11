#!/bin/bash
# GPU box: the top view's shape table alone (the loop of tools/gpu_round.sh): gpurun_out/<tag>_top_shapes_{kernels,steps}.txt -> tools/top_shapes_profile.py <tag>
tag=${1:-r05}; export TOPSHAPES_STEPS=240 TMPDIR=/tmp
: > gpurun_out/${tag}_top_shapes_kernels.txt; : > gpurun_out/${tag}_top_shapes_steps.txt
for shape in 8,8,32,256 8,16,32,512 16,16,32,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 8,16,24,512 16,16,20,256 9,9,32,256 9,12,32,256 12,12,32,256 8,8,16,256 8,8,64,256 24,24,32,256 32,32,32,1024 32,32,8,256; do
  tools/kprof.sh "top_$shape" tools/top_view_shapes.py $shape >> gpurun_out/${tag}_top_shapes_kernels.txt 2>&1 || echo "shape $shape failed"
  grep -h "^map" gpurun_out/kp_top_$shape.log >> gpurun_out/${tag}_top_shapes_steps.txt
done
# ... and once more WITHOUT the profiler (its tracing moves the cross-stream timings by a few us): section (c) of the profile
: > gpurun_out/${tag}_top_shapes_plain.txt
for shape in 8,8,32,256 8,16,32,512 16,16,32,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 8,16,24,512 16,16,20,256 9,9,32,256 9,12,32,256 12,12,32,256 8,8,16,256 8,8,64,256 24,24,32,256 32,32,32,1024 32,32,8,256; do
  timeout -k 10 120 python3 tools/top_view_shapes.py $shape 2>/dev/null | grep "^map" >> gpurun_out/${tag}_top_shapes_plain.txt
done
cut -c1-60,420-560 gpurun_out/${tag}_top_shapes_plain.txt
