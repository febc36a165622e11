#!/bin/bash
# Dev tool (GPU box): per-launch duration percentiles (tools/kdist.sh) of one tool run with the installed library and with
# each prebuilt variant (tools/_build/variants/*.so).   usage: tools/variant_dist.sh <script.py> [args...]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/variant_dist.txt
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  timeout -k 10 200 tools/kdist.sh "$(basename $v .so)" "$@" >> gpurun_out/variant_dist.txt 2>&1 || { tail -5 gpurun_out/variant_dist.txt; exit 1; }
done
grep -v "launch order" gpurun_out/variant_dist.txt
