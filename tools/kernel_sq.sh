#!/bin/bash
# Dev tool (GPU box): SQ counters of one kernel over a bench.py run — where its wavefronts' cycles go (instructions issued by
# kind, cycles waiting, cycles issuing).  Two rocprofv3 --pmc passes of eight counters each (counters in their own runs, with
# --kernel-trace only); per counter the mean over the kernel's dispatches (tools/pmc_summary.py).
#   usage: tools/kernel_sq.sh <kernel name substring, e.g. rcw_cast> <out file> [bench.py args...]
#          tools/kernel_sq.sh <kernel name substring> <out file> --tool tools/<script>.py [its args...]     (another program than bench.py)
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
k=$1; out=$2; shift 2
if [ "$1" = "--tool" ]; then prog=("$PWD/$2" "${@:3}"); else prog=("$PWD/bench.py" --no-cpu-baseline --steps 20 --warmup 2 "$@"); fi
R=$PWD; export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"
: > "$out"
n=0
for pass in "$P1" "$P2"; do
  n=$((n + 1)); rm -rf $R/gpurun_out/ksq_$n
  (cd /tmp && timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/ksq_$n -- python3 "${prog[@]}" > $R/gpurun_out/ksq_$n.log 2>&1) || { tail -5 gpurun_out/ksq_$n.log; exit 1; }
  for c in $pass; do python3 tools/pmc_summary.py gpurun_out/ksq_$n $c $k >> "$out"; done
  rm -rf $R/gpurun_out/ksq_$n
done
cat "$out"
