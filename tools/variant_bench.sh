#!/bin/bash
# Dev tool (GPU box): time bench.py with each prebuilt variant of the shared library (tools/_build/variants/*.so,
# built here with `make OUT=... EXTRA=-D...`) in place of the default one.   usage: tools/variant_bench.sh [bench args]
set -o pipefail
L=raycastworlds.jl_amd/lib/librcw_hip.so
cp $L /tmp/librcw_default.so
for v in default $(ls tools/_build/variants/*.so 2>/dev/null); do
  if [ "$v" = default ]; then cp /tmp/librcw_default.so $L; else cp "$v" $L; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python tools/bench_brief.py "$(basename $v .so)" || { cp /tmp/librcw_default.so $L; exit 1; }
done
cp /tmp/librcw_default.so $L
