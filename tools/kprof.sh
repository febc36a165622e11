#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel durations of one python tool run.   usage: tools/kprof.sh <tag> <script.py> [args...]
# Prints one line per rcw_ kernel: calls, average / min / median duration in us.  RCW_LIBRARY (a variant build) is passed through.
tag=$1; shift
export TMPDIR=/tmp; R=$PWD
rm -rf $R/gpurun_out/kp_$tag
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kp_$tag -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/kp_$tag.log 2>&1) || { echo "rocprofv3 failed: $tag"; tail -5 $R/gpurun_out/kp_$tag.log; exit 1; }
python3 - "$R/gpurun_out/kp_$tag" "$tag" <<'PY'
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
# per-launch durations too (the same run's kernel trace): the median says what a launch takes once the clocks are up — the
# first ten to twenty launches of a run are 5-15 % slower and pull the average of a 60-step run up
per = collections.defaultdict(list)
for t in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(t)):
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for r in csv.reader(open(f)):
    m = re.search(r"rcw_[a-z0-9_]+(<[^>]*>)?", r[0])
    if m and ("store" in r[0] or "fill" in r[0] or "draw" in r[0] or "top_view" in r[0] or "cast" in r[0] or "step" in r[0]):
        d = sorted(per.get(r[0], []))
        p50 = f"  p50 {d[len(d) // 2]:8.1f}" if d else ""
        print(f"{sys.argv[2]:28s} {m.group(0)[:44]:44s} calls {r[1]:>4s} avg {float(r[3]) / 1e3:8.1f} us  min {float(r[5]) / 1e3:8.1f}{p50}")
PY
rm -rf $R/gpurun_out/kp_$tag
