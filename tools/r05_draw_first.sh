#!/bin/bash
# Round 5, GPU box, development library: inside a step, the drawing on the handle's stream and the camera fill on the side stream
# (RCW_TOP_DRAW_FIRST=1) against the drawing on the side stream (0), over shapes the fused fill + draw launch does not take.
R=$PWD
export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out; out=gpurun_out/r05_draw_first.txt; : > $out
for spec in "24,24,32,256 -" "32,32,32,1024 -" "20,20,32,256 -" "22,22,32,256 -" "16,16,32,256 128" "16,16,32,256 300" "8,8,32,256 128" "8,8,32,256 300" "8,8,32,256 512" "8,16,32,512 128" "12,12,32,256 100" "8,8,13,256 128" "8,8,20,256 300"; do
  set -- $spec
  for f in 0 1; do
    hc=""; [ "$2" != "-" ] && hc="TOPSHAPES_HCAM=$2"
    echo -n "draw first $f H_cam ${2}: " >> $out
    env $hc RCW_TOP_DRAW_FIRST=$f TOPSHAPES_STEPS=120 timeout -k 10 120 python3 tools/top_view_shapes.py $1 2>&1 | grep "in a step" | cut -c1-400 >> $out
  done
done
cat $out
