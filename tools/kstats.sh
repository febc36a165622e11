#!/bin/bash
# usage: tools/kstats.sh <tag> [bench args...]  (on the GPU box) — rocprofv3 kernel stats of bench.py
tag=$1; shift
export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/$tag.log 2>&1)
grep -h "rcw_" $R/gpurun_out/$tag/*/*kernel_stats.csv | awk -F'","' '{n=split($1,a,"::"); printf "%-60s calls=%s avg_ns=%s min=%s max=%s\n", substr(a[n],1,58), $2, $4, $6, $7}'
grep -h '"value"' $R/gpurun_out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'launch_us', round(d['roofline']['launch_ms']*1e3,1), 'GB/s', round(d['roofline']['achieved']))"
