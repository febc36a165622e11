#!/bin/bash
# top view kernel iteration: its tests, then timing at a few persistent-grid sizes
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q -k "top_view or render_entry or float64 or discriminators or reference" > gpurun_out/b_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/b_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out ($rc): stopping"; exit $rc; fi
if [ $rc -ne 0 ]; then echo "tests failed: not timing"; exit $rc; fi
for g in 0 512 768 1024 1536 2048; do
  echo "== RCW_TOP_GRID=$g"
  if [ $g -eq 0 ]; then unset RCW_TOP_GRID; else export RCW_TOP_GRID=$g; fi
  timeout -k 10 120 python bench.py --top-view --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/b_top_g$g.json 2> gpurun_out/b_top_g$g.err || { echo "grid $g failed"; tail -3 gpurun_out/b_top_g$g.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('gpurun_out/b_top_g$g.json')); t=d['top_view']; print('top_view us', round(t['launch_ms']*1e3,1), 'GB/s', round(t['achieved']), 'frac', round(t['frac'],3), '| cast us', round(d['roofline']['whole_step']['cast_ms']*1e3,1), 'fill us', round(d['roofline']['launch_ms']*1e3,1))"
done
unset RCW_TOP_GRID
for w in cfg3 cfg4; do
  echo "== $w"
  timeout -k 10 120 python bench.py --workload $w --top-view --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b_top_$w.json 2> gpurun_out/b_top_$w.err || { echo "$w failed"; tail -3 gpurun_out/b_top_$w.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('gpurun_out/b_top_$w.json')); t=d['top_view']; print('top_view us', round(t['launch_ms']*1e3,1), 'GB/s', round(t['achieved']), 'frac', round(t['frac'],3))"
done
