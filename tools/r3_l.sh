#!/bin/bash
# round 3, GPU call L: bench lines per BASELINE config (+ the RLBase API loop at the headline), the two-piece step at cfg-3 / cfg-5
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --api rlbase --no-cpu-baseline > gpurun_out/r03_bench_cfg2_api.json 2> gpurun_out/r03_bench_cfg2_api.err; echo "cfg2 rc=$?"; cut -c1-260 gpurun_out/r03_bench_cfg2_api.json
for w in cfg3 cfg4 cfg5; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 10 --api rlbase > gpurun_out/r03_bench_$w.json 2> gpurun_out/r03_bench_$w.err; echo "$w rc=$?"; cut -c1-200 gpurun_out/r03_bench_$w.json
done
export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so
: > gpurun_out/r03_step_pieces.txt
for w in cfg3 cfg5 cfg2; do for pc in 1 2 1 2; do
  RCW_STEP_PIECES=$pc timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 10 2>/dev/null | python tools/bench_brief.py "$w pieces=$pc" >> gpurun_out/r03_step_pieces.txt
done; done
cat gpurun_out/r03_step_pieces.txt
