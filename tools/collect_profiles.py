#!/usr/bin/env python3
"""After `gpurun -- tools/gpu_round.sh <tag>`: copy the judged summaries from gpurun_out/ into profiles/.

    python tools/collect_profiles.py r02
"""
import csv
import json
import re
import shutil
import sys

tag = sys.argv[1]
shutil.copy(f"gpurun_out/{tag}_bench.json", f"profiles/{tag}_bench.json")
rows = list(csv.reader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
csv.writer(open(f"profiles/{tag}_kernel_stats.csv", "w")).writerows([rows[0]] + [[r[0][:160]] + r[1:] for r in rows[1:]])
w, f = open(f"gpurun_out/{tag}_write.txt").read(), open(f"gpurun_out/{tag}_fetch.txt").read()
W = float(re.search(r"rcw_fill256_kernel\s+WRITE_SIZE.*?mean=\s*([\d.]+)", w).group(1))
F = float(re.search(r"rcw_fill256_kernel\s+FETCH_SIZE.*?mean=\s*([\d.]+)", f).group(1))
alg = 4096 * 256 * 256 * 4 / 1024
open(f"profiles/{tag}_pmc_summary.txt", "w").write(
    "rocprofv3 PMC passes (separate runs, --pmc with --kernel-trace only), cfg2: 4096 agents x 256 columns, H_cam 256\n"
    "command: rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline\n"
    "Units: KiB per dispatch (sum over the 8 XCDs).  WRITE_SIZE is exact for 16-byte streaming stores; FETCH_SIZE on gfx950\n"
    "reports 1/2 of wide streaming reads and is uncalibrated for the fill kernel's narrow descriptor gathers.\n\n" + w + f +
    f"\nfill kernel: algorithmic {alg:,.0f} KiB per launch; WRITE_SIZE {W:,.1f} KiB (+{(W / alg - 1) * 100:.2f} %); FETCH_SIZE raw {F:,.1f} KiB\n")
json.dump({"workload": "cfg2", "batch": 4096, "kernel": "rcw_fill256_kernel", "write_bytes_per_launch": int(W * 1024),
           "fetch_bytes_per_launch_raw": int(F * 1024), "traffic_bytes_per_launch": int(W * 1024) + 2 * int(F * 1024),
           "note": "traffic = WRITE_SIZE + 2 x FETCH_SIZE (gfx950 FETCH correction, upper bound for narrow gathers)",
           "source": f"profiles/{tag}_pmc_summary.txt"}, open("profiles/pmc_traffic.json", "w"), indent=1)
b = json.load(open(f"profiles/{tag}_bench.json"))
print(tag, "value", round(b["value"]), "fill frac", round(b["roofline"]["frac"], 4), "step frac", round(b["roofline"]["whole_step"]["frac"], 4))
