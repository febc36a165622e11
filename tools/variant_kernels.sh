#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel averages of one tool run with the installed library and with each prebuilt variant
# (tools/_build/variants/*.so, built here with `make OUT=... EXTRA=-D...`, loaded by path through RCW_LIBRARY).
# usage: tools/variant_kernels.sh <script.py> [args...]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/variant_kernels.txt
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  timeout -k 10 200 tools/kprof.sh "$(basename $v .so)" "$@" >> gpurun_out/variant_kernels.txt 2>&1 || { tail -5 gpurun_out/variant_kernels.txt; exit 1; }
done
grep -v cast gpurun_out/variant_kernels.txt
