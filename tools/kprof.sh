#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel durations of one python tool run.   usage: tools/kprof.sh <tag> <script.py> [args...]
# Prints one line per rcw_ kernel: calls, average / min duration in us.  RCW_LIBRARY (a variant build) is passed through.
tag=$1; shift
export TMPDIR=/tmp; R=$PWD
rm -rf $R/gpurun_out/kp_$tag
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kp_$tag -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/kp_$tag.log 2>&1) || { echo "rocprofv3 failed: $tag"; tail -5 $R/gpurun_out/kp_$tag.log; exit 1; }
python3 - "$R/gpurun_out/kp_$tag" "$tag" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.reader(open(f)):
    m = re.search(r"rcw_[a-z0-9_]+(<[^>]*>)?", r[0])
    if m and ("store" in r[0] or "fill" in r[0] or "draw" in r[0] or "top_view" in r[0] or "cast" in r[0]):
        print(f"{sys.argv[2]:28s} {m.group(0)[:44]:44s} calls {r[1]:>4s} avg {float(r[3]) / 1e3:8.1f} us  min {float(r[5]) / 1e3:8.1f}")
PY
rm -rf $R/gpurun_out/kp_$tag
