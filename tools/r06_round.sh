#!/bin/bash
# GPU box (through gpurun), round 6, the FINAL binary: the bench lines, the rocprofv3 evidence for profiles/ and the shape tables.
# Everything lands under gpurun_out/r06_*; tools/r06_collect.py then writes the summaries into profiles/.   usage: tools/r06_round.sh [part]
#   part 1: bench lines of every workload in both step forms + kernel stats + PMC passes + the probes of the one-launch step
#   part 2: the top view (reference default act!, the shape table's (c) measure: what the top view adds to a step)
#   part 3: fuzz + soak
#   part 4: the GPU suite with durations (run LAST: profiles/r06_gpu_suite_durations.txt must come from the binary it names)
set -o pipefail
part=${1:-1}
R=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
stop_if_killed() { if [ "$1" -eq 124 ] || [ "$1" -eq 137 ]; then echo "step '$2' timed out ($1): stopping"; exit "$1"; fi; }
prof() {  # prof <outdir> <log> <rocprof args...> -- <bench args...>
  local out=$1 log=$2; shift 2
  local pre=() ; while [ "$1" != "--" ]; do pre+=("$1"); shift; done; shift
  rm -rf $R/gpurun_out/$out
  (cd /tmp && timeout -k 10 300 rocprofv3 "${pre[@]}" --output-format csv -d $R/gpurun_out/$out -- python3 $R/bench.py --no-cpu-baseline --traffic off "$@" > $R/gpurun_out/$log 2>&1)
  local rc=$?; stop_if_killed $rc "$out"; return $rc
}
if [ "$part" = 1 ]; then
  timeout -k 10 400 python bench.py --api rlbase > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; rc=$?; cut -c1-200 gpurun_out/r06_bench.json; stop_if_killed $rc bench
  timeout -k 10 400 python bench.py --step-form two-launches --no-cpu-baseline > gpurun_out/r06_bench_two_launches.json 2> /dev/null; rc=$?; stop_if_killed $rc bench2
  for w in cfg3 cfg4 cfg5; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r06_bench_$w.json 2> /dev/null; rc=$?; stop_if_killed $rc bench_$w
    timeout -k 10 300 python bench.py --workload $w --step-form two-launches --no-cpu-baseline --traffic off > gpurun_out/r06_bench_${w}_two_launches.json 2> /dev/null; rc=$?; stop_if_killed $rc bench2_$w
  done
  tools/r06_step_forms.sh > gpurun_out/r06_forms_final.txt 2>&1; cat gpurun_out/r06_forms_final.txt
  # kernel stats and HBM traffic of the headline command, both forms
  prof r06_stats r06_stats.log --kernel-trace --stats --
  prof r06_stats_two r06_stats_two.log --kernel-trace --stats -- --step-form two-launches
  prof r06_stats_cfg5 r06_stats_cfg5.log --kernel-trace --stats -- --workload cfg5 --steps 60 --warmup 5
  prof r06_pmc_write r06_pmc_write.log --pmc WRITE_SIZE --kernel-trace -- --steps 20 --warmup 2
  prof r06_pmc_fetch r06_pmc_fetch.log --pmc FETCH_SIZE --kernel-trace -- --steps 20 --warmup 2
  prof r06_pmc_sq r06_pmc_sq.log --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace -- --steps 20 --warmup 2
  for c in WRITE_SIZE; do python3 tools/pmc_summary.py gpurun_out/r06_pmc_write $c rcw_ > gpurun_out/r06_write.txt; done
  python3 tools/pmc_summary.py gpurun_out/r06_pmc_fetch FETCH_SIZE rcw_ > gpurun_out/r06_fetch.txt
  : > gpurun_out/r06_sq.txt; for c in SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY; do python3 tools/pmc_summary.py gpurun_out/r06_pmc_sq $c rcw_ >> gpurun_out/r06_sq.txt; done
  for t in r06_stats r06_stats_two r06_stats_cfg5; do cp gpurun_out/$t/*/*_kernel_stats.csv gpurun_out/${t}_kernel_stats.csv; done
  cat gpurun_out/r06_write.txt gpurun_out/r06_fetch.txt
  # what the casting half costs the launch (development build): the probes
  tools/r06_spec_halves.sh > gpurun_out/r06_spec_halves_final.txt 2>&1; cat gpurun_out/r06_spec_halves_final.txt
  echo "part 1 done"
fi
if [ "$part" = 2 ]; then
  timeout -k 10 300 python bench.py --no-cpu-baseline --top-view --steps 100 --warmup 10 > gpurun_out/r06_top_bench.json 2> gpurun_out/r06_top_bench.err; rc=$?; stop_if_killed $rc top_bench
  prof r06_top_stats r06_top_stats.log --kernel-trace --stats -- --top-view --steps 60 --warmup 5
  cp gpurun_out/r06_top_stats/*/*_kernel_stats.csv gpurun_out/r06_top_kernel_stats.csv
  export TOPSHAPES_STEPS=240
  : > gpurun_out/r06_top_shapes_plain.txt
  for shape in 8,8,32,256 8,16,32,512 16,16,32,256 8,8,10,256 8,8,12,256 8,8,13,256 8,8,20,256 8,8,24,256 8,16,24,512 16,16,20,256 9,9,32,256 9,12,32,256 12,12,32,256 8,8,16,256 8,8,64,256 24,24,32,256 32,32,32,1024 32,32,8,256; do
    timeout -k 10 120 python3 tools/top_view_shapes.py $shape 2>/dev/null | grep "^map" >> gpurun_out/r06_top_shapes_plain.txt
  done
  cut -c1-60,400-560 gpurun_out/r06_top_shapes_plain.txt
  HCAM_STEPS=200 timeout -k 10 300 python3 tools/hcam_bench.py > gpurun_out/r06_hcam_steps.txt 2>/dev/null; cat gpurun_out/r06_hcam_steps.txt
  echo "part 2 done"
fi
if [ "$part" = 3 ]; then
  out=gpurun_out/r06_fuzz.txt; : > $out
  run() { echo "== $*" >> $out; timeout -k 10 1000 python3 "$@" 2>&1 | tail -1 | cut -c1-700 >> $out; }
  run tools/fuzz_parity.py 500 61001
  run tools/fuzz_parity.py 600 61002 top
  run tools/fuzz_parity.py 600 61003 split
  run tools/fuzz_parity.py 600 61004 flat
  run tools/fuzz_parity.py 600 61005 step
  run tools/api_fuzz.py 120 61010 60
  run tools/api_fuzz.py 40 61011 60 sharded
  run tools/api_fuzz.py 30 61012 80 pairs
  echo "== tools/soak.py 100000" >> $out; timeout -k 10 600 python3 tools/soak.py 100000 2>&1 | tail -2 >> $out
  echo "== tools/soak.py 100000 one" >> $out; timeout -k 10 600 python3 tools/soak.py 100000 one 2>&1 | tail -2 >> $out
  cat $out
  echo "part 3 done"
fi
if [ "$part" = 4 ]; then
  timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r06_gpu_suite.log 2>&1; rc=$?; tail -32 gpurun_out/r06_gpu_suite.log; stop_if_killed $rc pytest
  make -s -C raycastworlds.jl_amd/csrc asm > /dev/null 2>&1
  tests/kernel_census.sh > gpurun_out/r06_kernel_census.log 2>&1; head -3 gpurun_out/census/summary.txt
  echo "part 4 done"
fi
