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
# ---- top view kernel (opt-in): kernel stats, bytes written, SQ counters
import os
if os.path.exists(f"gpurun_out/{tag}_top_kernel_stats.csv"):
    rows = list(csv.reader(open(f"gpurun_out/{tag}_top_kernel_stats.csv")))
    csv.writer(open(f"profiles/{tag}_top_view_kernel_stats.csv", "w")).writerows([rows[0]] + [[r[0][:160]] + r[1:] for r in rows[1:]])
    tw = open(f"gpurun_out/{tag}_top_write.txt").read()
    sq = open(f"gpurun_out/{tag}_top_sq.txt").read()
    def stat(name, table=rows):
        r = [r for r in table[1:] if name in r[0]]
        return (float(r[0][3]) / 1e3, int(r[0][1]), float(r[0][5]) / 1e3, float(r[0][6]) / 1e3) if r else None
    store, draw, fill = stat("rcw_top_store_kernel"), stat("rcw_top_draw_kernel"), stat("rcw_fill256_kernel")
    fused = stat("rcw_fill256_draw_kernel")             # round 4: the camera fill and the drawing in one launch
    TW = float(re.search(r"rcw_top_store_kernel\s+WRITE_SIZE.*?mean=\s*([\d.]+)", tw).group(1))
    m = re.search(r"rcw_top_draw_kernel\s+WRITE_SIZE.*?mean=\s*([\d.]+)", tw)
    DW = float(m.group(1)) if m else None
    m = re.search(r"rcw_fill256_draw_kernel\s+WRITE_SIZE.*?mean=\s*([\d.]+)", tw)
    FW = float(m.group(1)) if m else None
    fr = open(f"gpurun_out/{tag}_top_fetch.txt").read() if os.path.exists(f"gpurun_out/{tag}_top_fetch.txt") else ""
    ring = ""
    if os.path.exists(f"gpurun_out/{tag}_top_ring_kernel_stats.csv"):
        rr = stat("rcw_top_view_kernel", list(csv.reader(open(f"gpurun_out/{tag}_top_ring_kernel_stats.csv"))))
        if rr:
            ring = (f"\none-kernel form on the same workload (RCW_TOP_SPLIT=0: LDS bit planes, draw and store groups of one persistent kernel, what\n"
                    f"rcw_update_top_view alone and the geometries outside the two-kernel form's take): rcw_top_view_kernel avg {rr[0]:.1f} us "
                    f"= {1073741824 / rr[0] / 1e6 / 8 * 100:.1f} % of the HBM peak\n")
    side = ""
    if os.path.exists(f"gpurun_out/{tag}_top_side_kernel_stats.csv"):
        srows = list(csv.reader(open(f"gpurun_out/{tag}_top_side_kernel_stats.csv")))
        sd, sf, ss = stat("rcw_top_draw_kernel", srows), stat("rcw_fill256_kernel", srows), stat("rcw_top_store_kernel", srows)
        sb = json.loads(open(f"gpurun_out/{tag}_top_side_bench.json").read().strip().splitlines()[-1]) if os.path.exists(f"gpurun_out/{tag}_top_side_bench.json") else None
        if sd and sf and ss:
            side = (f"\nthe same over the side stream (rounds 2-3; development build, RCW_TOP_FUSED=0): rcw_top_draw_kernel avg {sd[0]:.1f} us beside "
                    f"rcw_fill256_kernel avg {sf[0]:.1f} us, store kernel {ss[0]:.1f} us"
                    + (f"; bench line {sb['ms_per_step'] * 1e3:.1f} us per step = {sb['value'] / 1e6:.2f} M env-steps/s" if sb else "") + "\n")
    bench_top = json.loads(open(f"gpurun_out/{tag}_top_bench.json").read().strip().splitlines()[-1]) if os.path.exists(f"gpurun_out/{tag}_top_bench.json") else None
    if fused:
        drawing = (f"fill + draw in one launch (rcw_fill256_draw_kernel: workgroups 0..255 the camera fill, one more per agent the drawing): avg {fused[0]:.1f} us "
                   f"(min {fused[2]:.1f}, max {fused[3]:.1f}) — the camera fill alone, in the headline run: see {tag}_kernel_stats.csv"
                   + (f"; WRITE_SIZE {FW:,.1f} KiB (frames 1,048,576 KiB + planes 32,768 KiB + codes)" if FW else "") + "\n")
    else:
        drawing = (f"draw kernel: avg {draw[0]:.1f} us (min {draw[2]:.1f}, max {draw[3]:.1f}), concurrent with rcw_fill256_kernel (avg {fill[0]:.1f} us in this run; "
                   f"alone, in the headline run: see {tag}_kernel_stats.csv); WRITE_SIZE {DW:,.1f} KiB (planes 32,768 KiB + codes)\n"
                   f"serial sum draw + store = {draw[0] + store[0]:.1f} us = {1073741824 / (draw[0] + store[0]) / 1e6 / 8 * 100:.1f} % if nothing ran beside the draw kernel\n")
    open(f"profiles/{tag}_top_view_summary.txt", "w").write(
        "update_top_view! (SR:446-483, opt-in), two-kernel form, cfg2 + pu_per_tu 32: 4096 agents x 256 x 256 px, 1 MI355X\n"
        "  drawing               rays -> Bresenham lines into an LDS bit plane -> plane (1/32 of the image) to HBM; one workgroup per agent, in the\n"
        "                        camera fill's launch (rcw_fill256_draw_kernel, round 4) or as rcw_top_draw_kernel on a side stream beside it\n"
        "  rcw_top_store_kernel  the fill kernel's moving window over the image: every pixel written once, 16 bytes a lane\n"
        "commands: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --top-view --steps 60 --warmup 5\n"
        "          rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --top-view --steps 20 --warmup 2   (and FETCH_SIZE)\n"
        "          rocprofv3 --pmc SQ_... --kernel-trace -- (same)\n\n"
        f"store kernel: avg {store[0]:.1f} us over {store[1]} dispatches (min {store[2]:.1f}, max {store[3]:.1f})\n"
        f"  algorithmic bytes per launch: 4 * 256 * 256 * 4096 = 1,073,741,824 B -> {1073741824 / store[0] / 1e6:.2f} TB/s = "
        f"{1073741824 / store[0] / 1e6 / 8 * 100:.1f} % of the 8 TB/s HBM peak\n"
        f"  WRITE_SIZE per launch: {TW:,.1f} KiB vs algorithmic 1,048,576 KiB (+{(TW / 1048576 - 1) * 100:.2f} %): every pixel is written once\n"
        + drawing
        + (f"bench line of the same workload: {bench_top['ms_per_step'] * 1e3:.1f} us per step = {bench_top['value'] / 1e6:.2f} M env-steps/s with both images rendered "
           f"(what the reference's act! does every step, SR:333-340)\n" if bench_top else "")
        + side + ring + "\n" + tw + fr + "\n" + sq)
    if bench_top:
        shutil.copy(f"gpurun_out/{tag}_top_bench.json", f"profiles/{tag}_bench_reference_default_act.json")
# ---- cast kernel at cfg-5: exec-masked march (shipped) vs ballot-bounded march (RCW_CAST_MARCH=ballot)
if os.path.exists(f"gpurun_out/{tag}_cfg5_exec_sq.txt"):
    out = ["rcw_cast_kernel_r3 (the round-3 kernel, which carries both marches; development build, RCW_CAST_KERNEL=r3; the shipped round-4 kernel runs the\n"
           "exec-masked march: profiles/r04_cast_kernel.txt) at cfg-5 (SingleRoom 32x32, 1024 columns, 8192 agents: rays up to 60 tile steps), 1 MI355X\n"
           "exec-masked march (shipped: per-lane `break`, the hardware exec mask retires finished lanes) vs ballot-bounded march\n"
           "(RCW_CAST_MARCH=ballot: wave-uniform loop bound via __ballot, finished lanes carried through selects)\n"
           "commands: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 3\n"
           "          rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 bench.py --no-cpu-baseline --workload cfg5 --steps 10 --warmup 2\n"
           "per dispatch, summed over the 8 XCDs; SQ_*_CYCLES count quad-cycles\n"]
    for march in ("exec", "ballot"):
        rows = list(csv.reader(open(f"gpurun_out/{tag}_cfg5_{march}_kernel_stats.csv")))
        cast = [r for r in rows[1:] if "rcw_cast_kernel" in r[0]][0]
        sq = open(f"gpurun_out/{tag}_cfg5_{march}_sq.txt").read()
        v = {m.group(1): float(m.group(2)) for m in re.finditer(r"rcw_cast_kernel\w*\s+(\w+)\s+dispatches=.*?mean=\s*([\d.]+)", sq)}
        lane = v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v.get("SQ_ACTIVE_INST_VALU", 1) * 64, 1)
        out.append(f"\n== {march}: kernel avg {float(cast[3]) / 1e3:.1f} us (min {float(cast[5]) / 1e3:.1f}, max {float(cast[6]) / 1e3:.1f}), "
                   f"{v.get('SQ_INSTS_VALU', 0) / v.get('SQ_WAVES', 1):.0f} VALU instructions per wavefront, "
                   f"waiting {v.get('SQ_WAIT_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1) * 100:.0f} % of wave cycles, "
                   f"active lanes per VALU instruction {lane * 100:.1f} %\n" + sq)
    open(f"profiles/{tag}_cast_march_cfg5.txt", "w").write("".join(out))
if os.path.exists(f"gpurun_out/{tag}_cast_table.txt"):
    open(f"profiles/{tag}_cast_table.txt", "w").write(
        "rcw_cast_kernel_r3 (the round-3 kernel, development build, RCW_CAST_KERNEL=r3), the heading's ray-table slice (5 N values): read directly from the L2-resident table by the lane\n"
        "that uses it (shipped, \"tablel2\") vs copied to LDS first and read back (RCW_CAST_TABLE=lds, \"tablelds\", the form\n"
        "north_star words).  rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --workload <cfg> --steps 30 --warmup 3\n\n"
        + open(f"gpurun_out/{tag}_cast_table.txt").read())
if os.path.exists("gpurun_out/rccl_world1.json"):
    shutil.copy("gpurun_out/rccl_world1.json", f"profiles/{tag}_rccl_world1.json")
b = json.load(open(f"profiles/{tag}_bench.json"))
print(tag, "value", round(b["value"]), "fill frac", round(b["roofline"]["frac"], 4), "step frac", round(b["roofline"]["whole_step"]["frac"], 4))
