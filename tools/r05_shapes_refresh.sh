#!/bin/bash
# GPU box: the top view's shape table alone (the loop of tools/gpu_round.sh): gpurun_out/<tag>_top_shapes_{kernels,steps}.txt -> tools/top_shapes_profile.py <tag>
tag=${1:-r05}; export TOPSHAPES_STEPS=240 TMPDIR=/tmp
: > gpurun_out/${tag}_top_shapes_kernels.txt; : > gpurun_out/${tag}_top_shapes_steps.txt
for shape in 8,8,32,256 8,16,32,512 16,16,32,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 8,16,24,512 16,16,20,256 9,9,32,256 9,12,32,256 12,12,32,256 8,8,16,256 8,8,64,256 24,24,32,256 32,32,32,1024 32,32,8,256; do
  tools/kprof.sh "top_$shape" tools/top_view_shapes.py $shape >> gpurun_out/${tag}_top_shapes_kernels.txt 2>&1 || echo "shape $shape failed"
  grep -h "^map" gpurun_out/kp_top_$shape.log >> gpurun_out/${tag}_top_shapes_steps.txt
done
cut -c1-60,200-420 gpurun_out/${tag}_top_shapes_steps.txt
