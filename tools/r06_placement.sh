#!/bin/bash
# Where the one-launch step's chunk loop lies in the code, against the launch's time (profiles/r06_step_forms.txt (5)).
#   tools/r06_placement.sh build   here: sixteen libraries under tools/_build/nops/ from rcw_cast.hip as it is, compiled WITHOUT -falign-loops,
#                                  with k = 0..15 s_nop at rcw_fill256_cast_kernel's entry (they differ in nothing but where everything lies);
#                                  tools/loop_lines.py says where the chunk loop of each landed
#   tools/r06_placement.sh run     GPU box: cfg3 (two passes) and cfg5 with each
set -e
R=$PWD; C=$R/raycastworlds.jl_amd/csrc; O=$R/raycastworlds.jl_amd/lib/obj/ship; D=$R/tools/_build/nops
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden"
if [ "$1" = build ]; then
  mkdir -p $D
  python3 - "$C/rcw_cast.hip" "$C/x_nops.hip" <<'PY'
import sys
s = open(sys.argv[1]).read()
a = "    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];\n#ifdef RCW_DEV_SWITCHES\n    if (p.spec_debug & ((int)blockIdx.x < fill_blocks ? 2 : 1)) return;"
assert s.count(a) == 1, s.count(a)           # rcw_fill256_cast_kernel
b = a.replace("lds[];\n", "lds[];\n#define RCW_STR2(x) #x\n#define RCW_STR(x) RCW_STR2(x)\n    asm volatile(\".rept \" RCW_STR(RCW_X_PADNOPS) \"\\n\\ts_nop 0\\n\\t.endr\");\n", 1)
open(sys.argv[2], "w").write(s.replace(a, b))
PY
  for base in 0 4 8 12; do
    for k in $base $((base+1)) $((base+2)) $((base+3)); do
      ( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -DRCW_X_PADNOPS=$k -c -o $D/x_nops_$k.o x_nops.hip 2>/dev/null &&
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $D/librcw_hip_nops_$k.so $O/rcw_api.o $D/x_nops_$k.o $O/rcw_fill.o $O/rcw_top_draw.o $O/rcw_top_store.o -ldl ) &
    done
    wait
  done
  rm -f $C/x_nops.hip $D/*.o
  for k in $(seq 0 15); do echo -n "k=$k: "; python3 tools/loop_lines.py $D/librcw_hip_nops_$k.so rcw_fill256_cast_kernelIfLb0ELb0ELb0 | sed 's/.*chunk loops/chunk loops/;s/; march.*//'; done
else
  run() { RCW_LIBRARY=$D/librcw_hip_nops_$2.so python bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 nops %2d' % $2, round(d['ms_per_step']*1e3,1), round(d['roofline']['launch_ms']*1e3,1))"; }
  for rep in 1 2; do for k in $(seq 0 15); do run cfg3 $k; done; done
  for k in $(seq 0 15); do run cfg5 $k; done
fi
