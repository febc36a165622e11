#!/bin/bash
# Round 5, GPU box: SQ counters of rcw_top_draw_kernel alone, round-5 body against the round-4 body (development library).
R=$PWD; export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out; : > gpurun_out/r05_draw_sq2.txt
for shape in ${SHAPES:-8,8,32,256 32,32,32,1024 8,8,10,256}; do
  for draw in r5 r4; do
    echo "== rcw_top_draw_kernel alone, $shape, body $draw" >> gpurun_out/r05_draw_sq2.txt
    RCW_TOP_DRAW=$draw tools/kernel_sq.sh rcw_top_draw gpurun_out/r05_sq_tmp.txt --tool tools/top_alone.py $shape two-kernels 20 > /dev/null 2>&1; cat gpurun_out/r05_sq_tmp.txt >> gpurun_out/r05_draw_sq2.txt
  done
done
cat gpurun_out/r05_draw_sq2.txt
