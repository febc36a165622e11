#!/bin/bash
# GPU box: library builds against each other on ONE box (the boxes differ by 1-2 % in what their HBM gives a fill): the whole step of
# cfg2 / cfg3 / cfg5, two passes, the libraries alternating.   usage: tools/r06_ab.sh <lib.so> <lib.so> ...   (TWO=1: the two-launch step)
for rep in 1 2; do
for lib in "$@"; do
  for w in ${WORKLOADS:-cfg2 cfg3 cfg5}; do
    RCW_LIBRARY=$PWD/$lib python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --traffic off ${TWO:+--step-form two-launches} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', '$w', round(d['ms_per_step']*1e3,1), round(d['roofline']['launch_ms']*1e3,1), d['roofline']['kernel'])"
  done
done
done
