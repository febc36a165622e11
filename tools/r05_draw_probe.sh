#!/bin/bash
# Round 5, GPU box: what bounds the top view's drawing, and rcw_update_top_view alone in both forms over shapes.
#   (1) tools/draw_trace.py — a draw workgroup's life in six stamps (measurement build with the development switches)
#   (2) tools/top_alone.py under rocprofv3 (tools/kprof.sh) — the kernels of the stand-alone call, one-kernel against two-kernel form
#   (3) SQ counters of rcw_top_draw_kernel / rcw_top_view_kernel (tools/kernel_sq.sh)
# Output: gpurun_out/r05_draw_*.txt
set -o pipefail
R=$PWD; DEV=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out
: > gpurun_out/r05_draw_trace.txt
for shape in 8,8,32,256 16,16,32,256 32,32,32,1024 8,8,10,256; do
  timeout -k 10 200 python3 tools/draw_trace.py $shape 6 >> gpurun_out/r05_draw_trace.txt 2>&1 || echo "draw_trace $shape failed" >> gpurun_out/r05_draw_trace.txt
done
cat gpurun_out/r05_draw_trace.txt
: > gpurun_out/r05_draw_alone.txt
export RCW_LIBRARY=$DEV
for shape in ${SHAPES:-8,8,32,256 8,16,32,512 16,16,32,256 24,24,32,256 32,32,32,1024 8,8,64,256 32,32,8,256 8,8,16,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 9,9,32,256 12,12,32,256}; do
  for form in one-kernel two-kernels; do
    tools/kprof.sh "alone_${shape}_$form" tools/top_alone.py $shape $form 40 >> gpurun_out/r05_draw_alone.txt 2>&1 || echo "alone $shape $form failed" >> gpurun_out/r05_draw_alone.txt
    grep -h "^alone" gpurun_out/kp_alone_${shape}_$form.log >> gpurun_out/r05_draw_alone.txt
  done
done
cat gpurun_out/r05_draw_alone.txt
: > gpurun_out/r05_draw_sq.txt
for shape in 8,8,32,256 32,32,32,1024; do
  echo "== rcw_top_draw_kernel alone, $shape" >> gpurun_out/r05_draw_sq.txt
  tools/kernel_sq.sh rcw_top_draw gpurun_out/r05_sq_tmp.txt --tool tools/top_alone.py $shape two-kernels 20 > /dev/null 2>&1; cat gpurun_out/r05_sq_tmp.txt >> gpurun_out/r05_draw_sq.txt
done
echo "== rcw_top_view_kernel (one-kernel form), 8,8,32,256" >> gpurun_out/r05_draw_sq.txt
tools/kernel_sq.sh rcw_top_view_kernel gpurun_out/r05_sq_tmp.txt --tool tools/top_alone.py 8,8,32,256 one-kernel 20 > /dev/null 2>&1; cat gpurun_out/r05_sq_tmp.txt >> gpurun_out/r05_draw_sq.txt
cat gpurun_out/r05_draw_sq.txt
