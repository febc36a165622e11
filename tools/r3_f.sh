#!/bin/bash
# round 3, GPU call F: per-call durations of the store kernels (kernel trace) to see their distribution
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp TOPSHAPES_STEPS=60; R=$PWD
for shape in 8,8,24,256 8,8,32,256; do
  rm -rf $R/gpurun_out/kt_$shape
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$shape -- python3 $R/tools/top_view_shapes.py $shape > $R/gpurun_out/kt_$shape.log 2>&1) || exit 1
  python3 - "$R/gpurun_out/kt_$shape" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = "store" if "top_store" in n else "draw" if "top_draw" in n else "fill" if "fill" in n else "cast" if "cast" in n else "ring" if "top_view" in n else "other"
    out.append((short, s, e))
t0 = out[0][1]
print("kernel start_us dur_us  (last 40 kernels)")
for k, s, e in out[-40:]:
    print(f"{k:6s} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}")
d = [(e - s) / 1e3 for k, s, e in out if k == "store"]
print("store durations:", " ".join(f"{x:.0f}" for x in d))
PY
  rm -rf $R/gpurun_out/kt_$shape
done
