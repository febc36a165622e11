#!/bin/bash
# Run on the GPU box (through gpurun) at the end of a round: GPU tests, the bench line, and the rocprofv3
# evidence for profiles/ (kernel stats + separate WRITE_SIZE / FETCH_SIZE passes).  Everything lands under
# gpurun_out/<tag>_*; tools/collect_profiles.py then copies the summaries into profiles/.
#   usage: tools/gpu_round.sh r02
set -o pipefail
tag=${1:-rXX}
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${tag}_pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; cut -c1-160 gpurun_out/${tag}_bench.json
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_fetch
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_write WRITE_SIZE | tee gpurun_out/${tag}_write.txt
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_fetch FETCH_SIZE | tee gpurun_out/${tag}_fetch.txt
cp gpurun_out/${tag}_stats/*/*_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
