#!/bin/bash
# GPU box, development build: timing probes of the one-launch step (RCW_SPEC_DEBUG bits: 1 = the casting workgroups return at once,
# 2 = the fill's, 4 = the casting half stores nothing, 8 = the turns' fans without their table loads, 16 = the current state's fan only;
# all but 0 give wrong frames), per workload, in ONE call so that the boxes' differences cancel.
export RCW_LIBRARY=$PWD/raycastworlds.jl_amd/lib/librcw_hip_dev.so
for w in ${WORKLOADS:-cfg2 cfg3 cfg5}; do
  for dbg in ${PROBES:-0 1 2 4 8 12 16 18}; do
    RCW_SPEC_DEBUG=$dbg timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w', 'RCW_SPEC_DEBUG=%-2d' % $dbg, f\"{d['ms_per_step']*1e3:8.1f} us/step  launch {r['launch_ms']*1e3:7.1f} us\")"
  done
  RCW_STEP_FORM=1 timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w', 'two launches     ', f\"{d['ms_per_step']*1e3:8.1f} us/step  launch {r['launch_ms']*1e3:7.1f} us  cast {r['whole_step']['cast_ms']*1e3:6.1f} us\")"
done
