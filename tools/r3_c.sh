#!/bin/bash
# round 3, GPU call C: compile-time timing experiments of the flat top store kernel (tools/_build/variants/exp*.so), rocprofv3 kernel times
set -o pipefail
mkdir -p gpurun_out
export TOPSHAPES_STEPS=60
: > gpurun_out/r3c.txt
for v in default $(ls tools/_build/variants/*.so); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  for shape in 8,8,24,256 8,8,13,256 8,8,32,256; do
    tools/kprof.sh "$(basename $v .so)_$shape" tools/top_view_shapes.py $shape >> gpurun_out/r3c.txt 2>&1 || exit 1
  done
done
grep -E "store|draw" gpurun_out/r3c.txt
