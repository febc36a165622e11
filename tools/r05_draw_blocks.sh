#!/bin/bash
# Round 5, GPU box: the draw kernel with 64 / 128 / 256 threads a workgroup (development library, RCW_TOP_DRAW_BLOCK), rcw_update_top_view alone
R=$PWD; export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out; : > gpurun_out/r05_draw_blocks.txt
for shape in ${SHAPES:-8,8,10,256 8,8,12,256 8,8,13,256 8,8,16,256 8,8,20,256 8,8,24,256 8,8,32,256 32,32,8,256 8,16,32,512}; do
  for blk in 64 128 256; do
    RCW_TOP_DRAW_BLOCK=$blk tools/kprof.sh "blk_${shape}_$blk" tools/top_alone.py $shape two-kernels 30 2>&1 | grep -E "top_draw_kernel|failed" | sed "s/^/block $blk /" >> gpurun_out/r05_draw_blocks.txt
  done
done
cat gpurun_out/r05_draw_blocks.txt
