#!/bin/bash
# Round 5, GPU box, development library: several draw workgroups an agent (RCW_TOP_PARTS = 1 .. 4) on batches of big images.
R=$PWD; mkdir -p gpurun_out; out=gpurun_out/r05_draw_parts.txt; : > $out
export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
for parts in 2 3 4; do RCW_TOP_PARTS=$parts timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -m gpu -k "top" -x 2>&1 | tail -1 | sed "s/^/parts $parts: /" >> $out; done
for spec in "32,32,32,1024 256" "32,32,32,1024 128" "32,32,32,1024 64" "24,24,32,256 455" "24,24,32,256 228" "24,24,32,256 114"; do
  set -- $spec
  for parts in 1 2 4; do
    echo -n "B $2 parts $parts: " >> $out
    TOPSHAPES_BATCH=$2 RCW_TOP_PARTS=$parts TOPSHAPES_STEPS=120 timeout -k 10 120 python3 tools/top_view_shapes.py $1 2>&1 | grep "in a step" | cut -c1-60,150-210,250-420 >> $out
    TOPSHAPES_BATCH=$2 RCW_TOP_PARTS=$parts tools/kprof.sh "pt_$1_$2_$parts" tools/top_view_shapes.py $1 2>&1 | grep "top_draw" >> $out
  done
done
cat $out
