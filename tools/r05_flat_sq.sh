#!/bin/bash
# Round 5, GPU box: SQ counters of the camera fill kernels at camera heights 256 (rcw_fill256_kernel), 300 / 100 / 40 (rcw_fill_flat_kernel, K = 2 / 4 / 8)
mkdir -p gpurun_out; : > gpurun_out/r05_flat_sq.txt
for hb in 256,4096 300,3495 100,10486 40,26214; do
  echo "== camera fill at H_cam,B = $hb" >> gpurun_out/r05_flat_sq.txt
  HCAM_STEPS=20 tools/kernel_sq.sh rcw_fill gpurun_out/r05_sq_tmp.txt --tool tools/hcam_bench.py $hb > /dev/null 2>&1; grep -v "rcw_fill256_draw" gpurun_out/r05_sq_tmp.txt >> gpurun_out/r05_flat_sq.txt
done
cat gpurun_out/r05_flat_sq.txt
