#!/bin/bash
# Dev tool (GPU box): per-call duration percentiles of the rcw_ kernels of one python tool run (rocprofv3 kernel trace).
# usage: tools/kdist.sh <tag> <script.py> [args...]
tag=$1; shift
export TMPDIR=/tmp; R=$PWD
rm -rf $R/gpurun_out/kd_$tag
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kd_$tag -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/kd_$tag.log 2>&1) || { echo "rocprofv3 failed: $tag"; tail -5 $R/gpurun_out/kd_$tag.log; exit 1; }
python3 - "$R/gpurun_out/kd_$tag" "$tag" <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"rcw_[a-z0-9_]+(<[^>]*>)?", r["Kernel_Name"])
    if m: d[m.group(0)[:40]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k, v in d.items():
    if "fill" not in k and "store" not in k: continue
    v.sort(); t = sorted(x[1] for x in v); n = len(t)
    q = lambda p: t[min(n - 1, int(p * n))]
    print(f"{sys.argv[2]:14s} {k:40s} n {n:4d} min {t[0]:7.1f} p10 {q(.1):7.1f} p50 {q(.5):7.1f} p90 {q(.9):7.1f} max {t[-1]:7.1f} mean {sum(t)/n:7.1f}")
    print("   in launch order:", " ".join(f"{x[1]:.0f}" for x in v[:60]))
PY
rm -rf $R/gpurun_out/kd_$tag
