#!/bin/bash
# Test infrastructure (GPU box, from the repository root: tests/kernel_census.sh; tools/gpu_round.sh runs it at the end of a round).
# Which kernel instantiations of the shipped library does the GPU suite launch?  rocprofv3 --kernel-trace over the in-process GPU test
# files and the fuzzers (each program directly after `--`; the tests that start child processes are left out: rank scripts, bench.py,
# the examples and the plain-C harness use the same kernels), kernel names against the .amdhsa_kernel list of `make asm`
# (raycastworlds.jl_amd/lib/asm/rcw_kernels.s: build it first).  Result: gpurun_out/census/summary.txt -> profiles/<tag>_kernel_census.txt.
set -o pipefail
R=$PWD; export TMPDIR=/tmp
rm -rf $R/gpurun_out/census; mkdir -p $R/gpurun_out/census
run() { local tag=$1; shift; (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/census/$tag -- "$@" > $R/gpurun_out/census/$tag.log 2>&1); echo "$tag rc=$?"; tail -1 $R/gpurun_out/census/$tag.log | cut -c1-150; }
run tests python3 -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_full_size.py $R/tests/test_reference_fixtures.py $R/tests/test_reference_invariants.py $R/tests/test_discriminators.py $R/tests/test_kernel_instantiations.py $R/tests/test_gpu_step_form.py -m gpu -q -p no:cacheprovider
run fuzz1 python3 $R/tools/fuzz_parity.py 120 2025
run fuzz2 python3 $R/tools/fuzz_parity.py 80 77 flat
run fuzz3 python3 $R/tools/fuzz_parity.py 80 78 split
run fuzz4 python3 $R/tools/api_fuzz.py 12 5 50
python3 - <<'PY'
import csv, glob, re, collections, os
calls = collections.Counter()
for f in glob.glob("gpurun_out/census/*/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        m = re.search(r"(rcw_[a-z0-9_]+(<[^>]*>)?)", name)
        if m:
            calls[m.group(1)] += int(r["Calls"])
open("gpurun_out/census/launched.txt", "w").write("".join(f"{k}\t{v}\n" for k, v in sorted(calls.items())))
print(len(calls), "distinct rcw_ kernels launched,", sum(calls.values()), "launches")
import subprocess
text = open("raycastworlds.jl_amd/lib/asm/rcw_kernels.s").read()
mangled = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M)
dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
shipped = sorted({re.sub(r"\(.*$", "", re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))) for n in dem})
never = [n for n in shipped if n not in calls]
with open("gpurun_out/census/summary.txt", "w") as f:
    f.write(f"{len(shipped)} kernel instantiations in the shipped build's ISA (make asm), {len(shipped) - len(never)} of them launched, {sum(calls.values())} launches in all\n")
    f.write("never launched: " + (", ".join(never) if never else "none") + "\n")
    f.write("launched by the development build only / not in the shipped ISA: " + ", ".join(sorted(n for n in calls if n not in shipped)) + "\n\n")
    for n in shipped:
        f.write(f"{calls.get(n, 0):8d}  {n}\n")
print(open("gpurun_out/census/summary.txt").read()[:1500])
PY
for t in tests fuzz1 fuzz2 fuzz3 fuzz4; do grep -v "amdgpu.ids\|rocprofv3\|^$" gpurun_out/census/$t.log | tail -4 | cut -c1-200; rm -rf gpurun_out/census/$t; done
du -sh gpurun_out | tail -1
