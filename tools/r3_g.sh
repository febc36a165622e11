#!/bin/bash
# round 3, GPU call G: SQ counters of the store kernels (flat vs unit kernel)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp TOPSHAPES_STEPS=12; R=$PWD
: > gpurun_out/r3g.txt
for shape in 8,8,24,256 8,8,32,256; do
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
    rm -rf $R/gpurun_out/pm
    (cd /tmp && timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pm -- python3 $R/tools/top_view_shapes.py $shape > $R/gpurun_out/pm.log 2>&1) || { tail -3 $R/gpurun_out/pm.log; continue; }
    echo "== $shape" >> gpurun_out/r3g.txt
    for c in $pass; do python3 tools/pmc_summary.py gpurun_out/pm $c rcw_top_store >> gpurun_out/r3g.txt; done
  done
done
rm -rf $R/gpurun_out/pm
cat gpurun_out/r3g.txt
