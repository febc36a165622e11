#!/bin/bash
# round 3: SQ counters of the cast kernel at cfg-2 (where does a wavefront's time go)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD; export TMPDIR=/tmp
rm -rf gpurun_out/r3z_sq gpurun_out/r3z_sq2
(cd /tmp && timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r3z_sq -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/gpurun_out/r3z_sq.log 2>&1) || { tail -5 gpurun_out/r3z_sq.log; exit 1; }
(cd /tmp && timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r3z_sq2 -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/gpurun_out/r3z_sq2.log 2>&1) || { tail -5 gpurun_out/r3z_sq2.log; exit 1; }
for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY; do python3 tools/pmc_summary.py gpurun_out/r3z_sq $c rcw_cast; done
for c in SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do python3 tools/pmc_summary.py gpurun_out/r3z_sq2 $c rcw_cast; done
