#!/bin/bash
# round 3, GPU call J: parity suite, then the cast kernel at cfg-5 / cfg-2 (kernel time + SQ counters)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3j_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3j_pytest.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert|Mismatch|FAILED" gpurun_out/r3j_pytest.log | head -30; exit 1; fi
export TMPDIR=/tmp; R=$PWD
: > gpurun_out/r3j.txt
for w in cfg5 cfg2 cfg3; do
  tools/kprof.sh "cast_$w" bench.py --no-cpu-baseline --workload $w --steps 40 --warmup 5 >> gpurun_out/r3j.txt 2>&1
done
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  rm -rf $R/gpurun_out/pm
  (cd /tmp && timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pm -- python3 $R/bench.py --no-cpu-baseline --workload cfg5 --steps 10 --warmup 2 > $R/gpurun_out/pm.log 2>&1) || { tail -3 $R/gpurun_out/pm.log; continue; }
  for c in $pass; do python3 tools/pmc_summary.py gpurun_out/pm $c rcw_cast >> gpurun_out/r3j.txt; done
done
rm -rf $R/gpurun_out/pm
cat gpurun_out/r3j.txt
