#!/bin/bash
# Dev tool (GPU box): time bench.py with each prebuilt variant of the shared library (tools/_build/variants/*.so,
# built here with `make OUT=... EXTRA=-D...`).  The installed library is never touched: the host layer loads the variant
# by path (RCW_LIBRARY, raycastworlds.jl_amd/_capi.py).   usage: tools/variant_bench.sh [bench args]
set -o pipefail
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then unset RCW_LIBRARY; else export RCW_LIBRARY="$PWD/$v"; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python tools/bench_brief.py "$(basename $v .so)" || exit 1
done
