#!/bin/bash
# Round 5, GPU box: the new draw body — parity first (every top-view test), then its timeline and the stand-alone call's kernels.
set -o pipefail
R=$PWD; DEV=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "top_view or top or fuzz or instantiations or captured or discriminators or hand_derived" > gpurun_out/r05_draw_check_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r05_draw_check_pytest.log
if [ $rc -ne 0 ]; then echo "parity failed: stopping"; exit $rc; fi
: > gpurun_out/r05_draw_trace2.txt
for shape in ${TRACE_SHAPES:-8,8,32,256 16,16,32,256 32,32,32,1024 8,8,10,256}; do
  timeout -k 10 200 python3 tools/draw_trace.py $shape 5 >> gpurun_out/r05_draw_trace2.txt 2>&1 || echo "draw_trace $shape failed" >> gpurun_out/r05_draw_trace2.txt
done
cat gpurun_out/r05_draw_trace2.txt
: > gpurun_out/r05_draw_alone2.txt
export RCW_LIBRARY=$DEV
for shape in ${SHAPES:-8,8,32,256 8,16,32,512 16,16,32,256 24,24,32,256 32,32,32,1024 32,32,8,256 8,8,16,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 9,9,32,256}; do
  for draw in r5 r4; do
    RCW_TOP_DRAW=$draw tools/kprof.sh "alone_${shape}_$draw" tools/top_alone.py $shape two-kernels 40 2>&1 | grep -E "draw_kernel|failed" | sed "s/^/$draw /" >> gpurun_out/r05_draw_alone2.txt
    grep -h "^alone" gpurun_out/kp_alone_${shape}_$draw.log | sed "s/^/$draw /" >> gpurun_out/r05_draw_alone2.txt
  done
done
cat gpurun_out/r05_draw_alone2.txt
