#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q -k "top_view or render_entry or float64 or fuzz" > gpurun_out/c_pytest.log 2>&1
rc=$?
tail -3 gpurun_out/c_pytest.log
if [ $rc -ne 0 ]; then echo "tests failed ($rc): not timing"; exit $rc; fi
run() {
  timeout -k 10 120 python bench.py --top-view --steps 40 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/c_top.json 2> gpurun_out/c_top.err || { echo "failed"; tail -3 gpurun_out/c_top.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('gpurun_out/c_top.json')); t=d['top_view']; print('top_view us', round(t['launch_ms']*1e3,1), 'GB/s', round(t['achieved']), 'frac', round(t['frac'],3))"
}
for dbg in 0 2; do echo "== RCW_TOP_DEBUG=$dbg"; RCW_TOP_DEBUG=$dbg run; done
for g in 768; do echo "== RCW_TOP_GRID=$g"; RCW_TOP_GRID=$g run; done
echo "== cfg3"; run --workload cfg3 --steps 10
