#!/bin/bash
# Round 5, GPU box, the final binary (round-5 draw kernel, stand-alone form rule, sampler status bit): differential fuzzing against the oracle.
mkdir -p gpurun_out; out=gpurun_out/r05_fuzz.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 1000 python3 "$@" 2>&1 | tail -1 | cut -c1-700 >> $out; }
run tools/fuzz_parity.py 500 51001
run tools/fuzz_parity.py 900 51002 top
run tools/fuzz_parity.py 900 51003 split
run tools/fuzz_parity.py 900 51004 flat
run tools/api_fuzz.py 100 51010 60
run tools/api_fuzz.py 40 51011 60 sharded
run tools/api_fuzz.py 30 51012 80 pairs
echo "== tools/soak.py 100000 top" >> $out; timeout -k 10 600 python3 tools/soak.py 100000 top 2>&1 | tail -2 >> $out
cat $out
