#!/bin/bash
# Dev tool (GPU box): one measurement with the installed library and with each prebuilt variant of it
# (tools/_build/variants/*.so, built here with `make -C raycastworlds.jl_amd/csrc OUT=$PWD/tools/_build/variants/<name>.so EXTRA=-D...`).
# The installed library is never touched: the host layer loads a variant by path (RCW_LIBRARY, raycastworlds.jl_amd/_capi.py).
#   tools/variants.sh bench   [bench.py args]      one line per variant: step and fill time by HIP events (tools/bench_brief.py)
#   tools/variants.sh stats   [bench.py args]      rocprofv3 kernel averages of bench.py per variant
#   tools/variants.sh kernels <script.py> [args]   rocprofv3 kernel averages of a tool run (tools/kprof.sh)
#   tools/variants.sh dist    <script.py> [args]   per-launch duration percentiles of a tool run (tools/kdist.sh)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mode=$1; shift
mkdir -p gpurun_out
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  n=$(basename "$v" .so)
  case $mode in
    bench)   timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python tools/bench_brief.py "$n" || exit 1 ;;
    stats)   echo "== $n"; timeout -k 10 300 tools/kstats.sh "vs_$n" "$@" || exit 1 ;;
    kernels) timeout -k 10 260 tools/kprof.sh "$n" "$@" || exit 1 ;;
    dist)    timeout -k 10 260 tools/kdist.sh "$n" "$@" || exit 1 ;;
    *) echo "usage: tools/variants.sh bench|stats|kernels|dist ..."; exit 2 ;;
  esac
done
