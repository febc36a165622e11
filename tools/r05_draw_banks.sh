#!/bin/bash
# Round 5, GPU box, development build: the draw kernel with wavefronts of one kind of line and bank-aware starts (RCW_TOP_DRAW=banks) against the shipped body.
R=$PWD; mkdir -p gpurun_out; out=gpurun_out/r05_draw_banks.txt; : > $out
export RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
RCW_TOP_DRAW=banks timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_kernel_instantiations.py -q -m gpu -k "top" -x 2>&1 | tail -2 >> $out
unset RCW_LIBRARY
for v in "" banks; do
  for shape in 32,32,32,1024 24,24,32,256 16,16,32,256 8,8,32,256; do
    echo "== body '$v' $shape" >> $out
    RCW_TOP_DRAW=$v timeout -k 10 200 python3 tools/draw_trace.py $shape 3 2>&1 | grep -A2 "^ launch" | tail -1 | cut -c1-250 >> $out
    RCW_LIBRARY=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so RCW_TOP_DRAW=$v tools/kprof.sh "bk_${shape}_$v" tools/top_alone.py $shape two-kernels 40 2>&1 | grep "top_draw" >> $out
  done
done
cat $out
