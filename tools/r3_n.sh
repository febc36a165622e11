#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3n.txt
for v in default $(ls tools/_build/variants/*.so 2>/dev/null) default; do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  tools/kprof.sh "$(basename $v .so)" bench.py --no-cpu-baseline --steps 200 --warmup 20 >> gpurun_out/r3n.txt 2>&1
  grep -h '"value"' gpurun_out/kp_$(basename $v .so).log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench: value %.2fM step %.1f us fill(events) %.1f us'%(d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['launch_ms']*1e3))" >> gpurun_out/r3n.txt
done
grep -E "fill256|bench" gpurun_out/r3n.txt
