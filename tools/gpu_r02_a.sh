#!/bin/bash
# Round-2 development run on the GPU box: GPU tests, then the top view kernel variants.
set -o pipefail
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/a_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/a_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out ($rc): stopping"; exit $rc; fi
for v in 0 1 2 3 5; do
  echo "== RCW_TOP_VARIANT=$v"
  RCW_TOP_VARIANT=$v timeout -k 10 120 python bench.py --top-view --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/a_top_v$v.json 2> gpurun_out/a_top_v$v.err || { echo "variant $v failed"; tail -3 gpurun_out/a_top_v$v.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('gpurun_out/a_top_v$v.json')); t=d['top_view']; print('top_view us', round(t['launch_ms']*1e3,1), 'GB/s', round(t['achieved']), 'frac', round(t['frac'],3), '| cast us', round(d['roofline']['whole_step']['cast_ms']*1e3,1), 'fill us', round(d['roofline']['launch_ms']*1e3,1))"
done
echo "== in-place kernel"
RCW_TOP_INPLACE=1 timeout -k 10 120 python bench.py --top-view --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/a_top_inplace.json 2> gpurun_out/a_top_inplace.err
python3 -c "import json,sys; d=json.load(open('gpurun_out/a_top_inplace.json')); t=d['top_view']; print('top_view us', round(t['launch_ms']*1e3,1), 'GB/s', round(t['achieved']), 'frac', round(t['frac'],3))"
