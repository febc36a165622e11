#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel stats of bench.py.   usage: tools/kstats.sh <tag> [bench args...]
# Prints one line per rcw_ kernel (calls, average / min / max duration in us) and the bench line's headline figures.
tag=$1; shift
export TMPDIR=/tmp; R=$PWD
rm -rf $R/gpurun_out/$tag
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$tag.log 2>&1) || { echo "rocprofv3 failed: $tag"; tail -5 $R/gpurun_out/$tag.log; exit 1; }
python3 - "$R/gpurun_out/$tag" "$R/gpurun_out/$tag.log" <<'PY'
import csv, glob, json, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.reader(open(f)):
    m = re.search(r"rcw_[a-z0-9_]+(<[^>]*>)?", r[0])
    if m:
        print(f"  {m.group(0)[:46]:46s} calls {r[1]:>5s}  avg {float(r[3]) / 1e3:9.2f} us  min {float(r[5]) / 1e3:9.2f}  max {float(r[6]) / 1e3:9.2f}")
for line in open(sys.argv[2]):
    if line.startswith("{") and '"value"' in line:
        d = json.loads(line)
        print(f"  bench: {d['value'] / 1e6:.2f} M env-steps/s, step {d['ms_per_step'] * 1e3:.1f} us (host clock), fill by events {d['roofline']['launch_ms'] * 1e3:.1f} us, "
              f"cast by events {d['roofline']['whole_step']['cast_ms'] * 1e3:.1f} us")
PY
