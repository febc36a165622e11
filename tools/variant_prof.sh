#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel stats of bench.py with each prebuilt variant of the shared library (see
# variant_bench.sh; the variant is loaded by path through RCW_LIBRARY, the installed library stays as it is).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$R/$v"; fi
  n=$(basename $v .so); rm -rf $R/gpurun_out/vp_$n
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vp_$n -- python3 $R/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1) || { echo "rocprofv3 failed for $n"; continue; }
  echo "== $n"; python3 - "$R/gpurun_out/vp_$n" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.reader(open(f)):
    if "rcw_" in r[0]:
        import re
        print(f"  {re.search(r'rcw_[a-z0-9_]+', r[0]).group(0):28s} calls {r[1]:>4s} avg {float(r[3]) / 1e3:8.1f} us")
PY
done
