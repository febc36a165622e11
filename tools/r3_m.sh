#!/bin/bash
# round 3, GPU call M: differential fuzzing of the new kernels against the oracle
mkdir -p gpurun_out
for args in "600 11 flat" "300 12 top" "300 13 split" "400 14"; do
  timeout -k 10 900 python -u tools/fuzz_parity.py $args > gpurun_out/r3m_fuzz_$(echo $args | tr ' ' '_').log 2>&1
  echo "fuzz $args: rc=$? $(tail -1 gpurun_out/r3m_fuzz_$(echo $args | tr ' ' '_').log)"
  grep FAILED -A1 gpurun_out/r3m_fuzz_$(echo $args | tr ' ' '_').log | head -12
done
