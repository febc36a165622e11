#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output for the fill kernel into profiles/ (HBM bytes per launch).

usage: tools/pmc_summary.py <dir with *_counter_collection.csv> <COUNTER> [kernel substring]
Prints mean counter value per dispatch of the matching kernel.  FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x (MI355X_MICROARCH.md §HBM).
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d, counter = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else "rcw_"
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(list)
    for f in files:
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or pat not in r["Kernel_Name"]:
                continue
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for k, v in per_dispatch.items():
            acc[names[k]].append(v)
    for name, vals in sorted(acc.items()):
        import re
        m = re.search(r"rcw_\w+", name)
        short = m.group(0) if m else name[:28]
        print(f"{short:28s} {counter:11s} dispatches={len(vals):4d} mean={sum(vals)/len(vals):16.1f} min={min(vals):16.1f} max={max(vals):16.1f}")


if __name__ == "__main__":
    main()
