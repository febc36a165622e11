#!/bin/bash
# record_stream restored in act_ / expand_columns / gather: first the one test that exercises it, alone (an abort must not
# take the suite with it), then the whole suite
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dropped_right_after or rlbase_verbs" > gpurun_out/r3k_one.log 2>&1; rc=$?
tail -3 gpurun_out/r3k_one.log; echo "single test exit code $rc"
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r3k_one.log; exit 1; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3k_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r3k_pytest.log
