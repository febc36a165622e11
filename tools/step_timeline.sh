#!/bin/bash
# Dev tool (GPU box): the kernels of a step with the top view on a common time axis (rocprofv3 --kernel-trace of tools/top_view_shapes.py):
# per step, start and end of the cast, draw, fill and store kernels relative to the cast kernel's start — median over the run's steps.
#   usage: tools/step_timeline.sh H,W,pu,N [steps=120]
shape=$1; export TOPSHAPES_STEPS=${2:-120}
export TMPDIR=/tmp; R=$PWD; out=$R/gpurun_out/tl_$shape
rm -rf $out
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/top_view_shapes.py $shape > $out.log 2>&1) || { echo "rocprofv3 failed"; tail -5 $out.log; exit 1; }
python3 - $out $shape <<'PY'
import csv, glob, statistics, sys
rows = []
for t in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(t)):
        n = r["Kernel_Name"]
        k = "cast" if "rcw_cast" in n else "draw" if "top_draw" in n else "fill+draw" if "fill256_draw" in n else "fill" if "rcw_fill" in n else "store" if "top_store" in n else None
        if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
steps, cur = [], None
for s, e, k in rows:
    if k == "cast":
        if cur and len(cur) >= 3: steps.append(cur)
        cur = {"cast": (s, e)}
    elif cur is not None and k not in cur:
        cur[k] = (s, e)
steps = [st for st in steps if "store" in st][10:]          # (the first steps: clocks)
print(f"{sys.argv[2]}: {len(steps)} steps; microseconds from the cast kernel's start (median)")
for k in ("cast", "draw", "fill", "fill+draw", "store"):
    v = [st[k] for st in steps if k in st]
    if not v: continue
    t0 = [st["cast"][0] for st in steps if k in st]
    a = statistics.median((x[0] - z) / 1e3 for x, z in zip(v, t0)); b = statistics.median((x[1] - z) / 1e3 for x, z in zip(v, t0))
    print(f"  {k:10s} start {a:7.1f}  end {b:7.1f}  ({b - a:6.1f} us)")
both = [st for st in steps if "draw" in st and "fill" in st]
if both:
    print("  store start - max(draw end, fill end): %.1f us (median)" % statistics.median((st["store"][0] - max(st["draw"][1], st["fill"][1])) / 1e3 for st in both))
    print("  draw start - cast end: %.1f us; fill start - cast end: %.1f us" % (statistics.median((st["draw"][0] - st["cast"][1]) / 1e3 for st in both), statistics.median((st["fill"][0] - st["cast"][1]) / 1e3 for st in both)))
PY
rm -rf $out
