#!/bin/bash
# round 3: a longer differential fuzz (new seeds) in all four modes + the soak
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/r3_fuzz.txt
for mode in "" top split flat; do
  echo "== mode '${mode:-camera}'" >> gpurun_out/r3_fuzz.txt
  timeout -k 10 ${FUZZ_T:-260} python tools/fuzz_parity.py ${FUZZ_N:-350} $((${FUZZ_SEED:-7000} + ${#mode})) $mode 2>&1 | grep -v "^config" >> gpurun_out/r3_fuzz.txt; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "fuzz '$mode' timed out" | tee -a gpurun_out/r3_fuzz.txt; exit $rc; fi
done
cat gpurun_out/r3_fuzz.txt | tail -20
