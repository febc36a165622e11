#!/bin/bash
# round 3, GPU call D: top view parity, then rocprofv3 kernel times of the flat kernels on their shapes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "top_view or camera_heights" > gpurun_out/r3d_pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r3d_pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
export TOPSHAPES_STEPS=60
: > gpurun_out/r3d.txt
for shape in 8,8,24,256 8,8,13,256 9,9,32,256 8,8,10,256 8,8,12,256 8,8,20,256 16,16,20,256; do
  tools/kprof.sh "top_$shape" tools/top_view_shapes.py $shape >> gpurun_out/r3d.txt 2>&1 || exit 1
done
tools/kprof.sh hcam tools/hcam_bench.py >> gpurun_out/r3d.txt 2>&1
grep -E "store|draw|fill_flat|fill_window|fill256" gpurun_out/r3d.txt
