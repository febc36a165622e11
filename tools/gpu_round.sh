#!/bin/bash
# Run on the GPU box (through gpurun) at the end of a round: GPU tests, the bench line, and the rocprofv3
# evidence for profiles/ (kernel stats + separate WRITE_SIZE / FETCH_SIZE passes; top view; cast kernel at cfg-5).
# Everything lands under gpurun_out/<tag>_*; tools/collect_profiles.py then copies the summaries into profiles/.
#   usage: tools/gpu_round.sh r02 [notests]
set -o pipefail
tag=${1:-rXX}
R=$PWD
DEVLIB=$R/raycastworlds.jl_amd/lib/librcw_hip_dev.so
mkdir -p gpurun_out
export TMPDIR=/tmp
stop_if_killed() { if [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; then echo "step '$2' timed out ($1): stopping"; exit "$1"; fi; }

if [ "$2" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; rc=$?; tail -2 gpurun_out/${tag}_pytest_gpu.log; stop_if_killed $rc pytest
fi
timeout -k 10 300 python bench.py --api rlbase > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; rc=$?; cut -c1-160 gpurun_out/${tag}_bench.json; stop_if_killed $rc bench
(cd $R && timeout -k 10 300 python bench.py --no-cpu-baseline --top-view --steps 100 --warmup 10 > gpurun_out/${tag}_top_bench.json 2> gpurun_out/${tag}_top_bench.err); rc=$?; stop_if_killed $rc top_bench
rm -rf gpurun_out/${tag}_top_pmc_fetch gpurun_out/${tag}_top_side_stats gpurun_out/${tag}_top_ring_stats gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_top_stats gpurun_out/${tag}_top_pmc_write gpurun_out/${tag}_cfg5_*
cd /tmp
prof() {  # prof <outdir> <log> <rocprof args...> -- <bench args...>
  local out=$1 log=$2; shift 2
  local pre=() ; while [ "$1" != "--" ]; do pre+=("$1"); shift; done; shift
  timeout -k 10 300 rocprofv3 "${pre[@]}" --output-format csv -d $R/gpurun_out/$out -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$log 2>&1
  local rc=$?; stop_if_killed $rc "$out"; return $rc
}
# --- headline workload (cfg2): kernel stats + HBM traffic passes
prof ${tag}_stats ${tag}_stats.log --kernel-trace --stats --
prof ${tag}_pmc_write ${tag}_pmc_write.log --pmc WRITE_SIZE --kernel-trace -- --steps 20 --warmup 2
prof ${tag}_pmc_fetch ${tag}_pmc_fetch.log --pmc FETCH_SIZE --kernel-trace -- --steps 20 --warmup 2
# --- top view kernel (opt-in): kernel stats + bytes written
prof ${tag}_top_stats ${tag}_top_stats.log --kernel-trace --stats -- --top-view --steps 60 --warmup 5
prof ${tag}_top_pmc_write ${tag}_top_pmc_write.log --pmc WRITE_SIZE --kernel-trace -- --top-view --steps 20 --warmup 2
prof ${tag}_top_pmc_fetch ${tag}_top_pmc_fetch.log --pmc FETCH_SIZE --kernel-trace -- --top-view --steps 20 --warmup 2
RCW_LIBRARY=$DEVLIB RCW_TOP_FUSED=0 prof ${tag}_top_side_stats ${tag}_top_side_stats.log --kernel-trace --stats -- --top-view --steps 60 --warmup 5
(cd $R && RCW_LIBRARY=$DEVLIB RCW_TOP_FUSED=0 timeout -k 10 300 python bench.py --no-cpu-baseline --top-view --steps 100 --warmup 10 > gpurun_out/${tag}_top_side_bench.json 2> /dev/null)
RCW_LIBRARY=$DEVLIB RCW_TOP_SPLIT=0 prof ${tag}_top_ring_stats ${tag}_top_ring_stats.log --kernel-trace --stats -- --top-view --steps 60 --warmup 5
prof ${tag}_top_pmc_sq ${tag}_top_pmc_sq.log --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -- --top-view --steps 20 --warmup 2
# --- cast kernel at cfg-5 (32x32 map, 1024 columns: 60-step rays), exec-masked march vs ballot-bounded march
export RCW_LIBRARY=$DEVLIB RCW_CAST_KERNEL=r3   # (the measured-and-rejected variants live in the development build only, in the round-3 kernel: both legs from it)
for march in exec ballot; do
  export RCW_CAST_MARCH=$march
  prof ${tag}_cfg5_${march}_stats ${tag}_cfg5_${march}_stats.log --kernel-trace --stats -- --workload cfg5 --steps 30 --warmup 3
  prof ${tag}_cfg5_${march}_sq ${tag}_cfg5_${march}_sq.log --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -- --workload cfg5 --steps 10 --warmup 2
  prof ${tag}_cfg5_${march}_lanes ${tag}_cfg5_${march}_lanes.log --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU --kernel-trace -- --workload cfg5 --steps 10 --warmup 2
done
unset RCW_CAST_MARCH
# --- cast kernel with the heading's table slice staged in LDS first (RCW_CAST_TABLE=lds) vs the direct L2 read
for w in cfg2 cfg5; do
  RCW_CAST_TABLE=lds prof ${tag}_${w}_tablelds_stats ${tag}_${w}_tablelds_stats.log --kernel-trace --stats -- --workload $w --steps 30 --warmup 3
  prof ${tag}_${w}_tablel2_stats ${tag}_${w}_tablel2_stats.log --kernel-trace --stats -- --workload $w --steps 30 --warmup 3
done
unset RCW_LIBRARY RCW_CAST_KERNEL
cd $R
# --- the flat kernels (any camera height / any top-view pixel scale): per-kernel times by rocprofv3 on their shapes, bytes written
export TOPSHAPES_STEPS=240
: > gpurun_out/${tag}_top_shapes_kernels.txt; : > gpurun_out/${tag}_top_shapes_steps.txt
for shape in 8,8,32,256 8,16,32,512 16,16,32,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 8,16,24,512 16,16,20,256 9,9,32,256 9,12,32,256 12,12,32,256 8,8,16,256 8,8,64,256 24,24,32,256 32,32,32,1024 32,32,8,256; do
  tools/kprof.sh "top_$shape" tools/top_view_shapes.py $shape >> gpurun_out/${tag}_top_shapes_kernels.txt 2>&1 || echo "shape $shape failed"
  grep -h "^map" gpurun_out/kp_top_$shape.log >> gpurun_out/${tag}_top_shapes_steps.txt
done
HCAM_STEPS=200 tools/kprof.sh hcam tools/hcam_bench.py > gpurun_out/${tag}_hcam_kernels.txt 2>&1; grep -h "^H_cam" gpurun_out/kp_hcam.log > gpurun_out/${tag}_hcam_steps.txt
export TMPDIR=/tmp TOPSHAPES_STEPS=12
for what in "top_view_shapes.py 8,8,24,256" "top_view_shapes.py 8,8,13,256" "hcam_bench.py 100,10486 250,4194"; do
  n=$(echo $what | tr ' ,.' '___'); rm -rf $R/gpurun_out/${tag}_flat_write_$n
  (cd /tmp && timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_flat_write_$n -- python3 $R/tools/$what > $R/gpurun_out/${tag}_flat_write_$n.log 2>&1)
  python3 tools/pmc_summary.py gpurun_out/${tag}_flat_write_$n WRITE_SIZE rcw_ | grep -E "flat|fill256|top_store" >> gpurun_out/${tag}_flat_write.txt
done
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_write WRITE_SIZE | tee gpurun_out/${tag}_write.txt
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_fetch FETCH_SIZE | tee gpurun_out/${tag}_fetch.txt
python3 tools/pmc_summary.py gpurun_out/${tag}_top_pmc_write WRITE_SIZE | tee gpurun_out/${tag}_top_write.txt
python3 tools/pmc_summary.py gpurun_out/${tag}_top_pmc_fetch FETCH_SIZE rcw_top | tee gpurun_out/${tag}_top_fetch.txt
cp gpurun_out/${tag}_top_ring_stats/*/*_kernel_stats.csv gpurun_out/${tag}_top_ring_kernel_stats.csv
cp gpurun_out/${tag}_top_side_stats/*/*_kernel_stats.csv gpurun_out/${tag}_top_side_kernel_stats.csv
cp gpurun_out/${tag}_stats/*/*_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
cp gpurun_out/${tag}_top_stats/*/*_kernel_stats.csv gpurun_out/${tag}_top_kernel_stats.csv
for march in exec ballot; do cp gpurun_out/${tag}_cfg5_${march}_stats/*/*_kernel_stats.csv gpurun_out/${tag}_cfg5_${march}_kernel_stats.csv; done
: > gpurun_out/${tag}_top_sq.txt
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT; do python3 tools/pmc_summary.py gpurun_out/${tag}_top_pmc_sq $c rcw_top >> gpurun_out/${tag}_top_sq.txt; done
for march in exec ballot; do
  : > gpurun_out/${tag}_cfg5_${march}_sq.txt
  for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE; do python3 tools/pmc_summary.py gpurun_out/${tag}_cfg5_${march}_sq $c rcw_cast >> gpurun_out/${tag}_cfg5_${march}_sq.txt; done
  for c in SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU; do python3 tools/pmc_summary.py gpurun_out/${tag}_cfg5_${march}_lanes $c rcw_cast >> gpurun_out/${tag}_cfg5_${march}_sq.txt; done
done
for w in cfg2 cfg5; do for v in tablelds tablel2; do echo "$w $v calls,total_ns,avg_ns,pct,min_ns,max_ns,stddev: $(grep -h rcw_cast_kernel gpurun_out/${tag}_${w}_${v}_stats/*/*_kernel_stats.csv | sed 's/.*)",//' | tail -1)"; done; done | tee gpurun_out/${tag}_cast_table.txt
grep -h "rcw_" gpurun_out/${tag}_cfg5_*_kernel_stats.csv | cut -c1-200 | head -8
echo "round script done"
# --- which kernel instantiations of the shipped build the GPU suite launches (tests/kernel_census.sh -> gpurun_out/census/summary.txt)
make -s -C raycastworlds.jl_amd/csrc asm > /dev/null 2>&1
tests/kernel_census.sh > gpurun_out/${tag}_kernel_census.log 2>&1; head -3 gpurun_out/census/summary.txt
echo "round script done (census)"
