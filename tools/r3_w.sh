#!/bin/bash
# round 3: cast kernel block size (development library's RCW_CAST_BLOCK) at cfg-2 / cfg-3 / cfg-5
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/r3_w.txt
export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so
for b in 256 192 128 64; do
  for w in cfg2 cfg3 cfg5; do
    RCW_CAST_BLOCK=$b timeout -k 10 200 tools/kprof.sh "block$b-$w" bench.py --no-cpu-baseline --workload $w --steps 40 --warmup 5 2>&1 | grep "cast" >> gpurun_out/r3_w.txt
  done
done
cat gpurun_out/r3_w.txt
