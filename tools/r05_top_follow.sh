#!/bin/bash
# Round 5, GPU box, development library: the experiment RCW_TOP_FOLLOW (the top view's store kernel following its draw kernel through
# counters in memory instead of an event; docs/experiments.md) — pixels under stress, then in-step and stand-alone times with it off / on.
R=$PWD
export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out; out=gpurun_out/r05_top_follow.txt; : > $out
RCW_TOP_FOLLOW=3 timeout -k 10 120 python3 tools/top_follow_stress.py both 2>&1 | grep -c "mismatching pixels 0" | sed 's/^/stress: comparisons without a mismatching pixel: /' >> $out
RCW_TOP_FOLLOW=3 timeout -k 10 120 python3 tools/top_follow_stress.py both 2>&1 | grep "mismatching" | grep -vc "pixels 0" | sed 's/^/stress: comparisons WITH mismatching pixels: /' >> $out
for shape in 8,8,32,256 8,16,32,512 16,16,32,256 24,24,32,256 32,32,32,1024 8,8,10,256 8,8,13,256 8,8,16,256 8,8,20,256 12,12,32,256; do
  for f in 0 3; do
    echo -n "follow $f: " >> $out; RCW_TOP_ALONE_SPLIT=1 RCW_TOP_FOLLOW=$f TOPSHAPES_STEPS=120 timeout -k 10 120 python3 tools/top_view_shapes.py $shape 2>&1 | grep "in a step" | cut -c1-240 >> $out
  done
done
cat $out
