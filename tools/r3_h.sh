#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3h.txt
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  tools/kprof.sh "$(basename $v .so)" tools/hcam_bench.py 100,10486 250,4194 300,3495 >> gpurun_out/r3h.txt 2>&1
done
grep -E "fill_flat" gpurun_out/r3h.txt
