#!/bin/bash
# GPU box: the bench line of every BASELINE workload in both forms of the step (rcw_set_step_form), 200 steps each,
# without the CPU baseline and the counter passes: gpurun_out/r06_forms_<workload>_<form>.json
set -o pipefail
mkdir -p gpurun_out
for w in cfg2 cfg3 cfg4 cfg5; do
  for f in one-launch two-launches; do
    timeout -k 10 200 python bench.py --workload $w --step-form $f --steps 200 --warmup 20 --no-cpu-baseline --traffic off > gpurun_out/r06_forms_${w}_${f}.json 2> gpurun_out/r06_forms_${w}_${f}.err || { echo "$w $f failed"; exit 1; }
    python - "$w" "$f" <<'PY'
import json,sys
w,f=sys.argv[1:3]
d=json.load(open(f"gpurun_out/r06_forms_{w}_{f}.json"))
r=d["roofline"]
print(f"{w} {f:13s} {d['value']/1e6:7.2f} M env-steps/s  {d['ms_per_step']*1e3:8.1f} us/step  launch {r['launch_ms']*1e3:7.1f} us frac {r['frac']:.3f}  cast {r['whole_step']['cast_ms']*1e3:6.1f} us  whole step {r['whole_step']['frac']:.3f}  [{r['kernel']}]")
PY
  done
done
