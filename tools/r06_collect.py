#!/usr/bin/env python3
"""After `gpurun -- tools/r06_round.sh <part>`: copy / condense the judged summaries from gpurun_out/ into profiles/ (round 6).

    python tools/r06_collect.py"""
import csv
import json
import os
import re
import shutil

G, P = "gpurun_out", "profiles"


def line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def stats(path):
    rows = list(csv.reader(open(path)))
    return [rows[0]] + [[r[0][:170]] + r[1:] for r in rows[1:]]


def kernel(rows, name):
    r = [r for r in rows[1:] if name in r[0]]
    return (int(r[0][1]), float(r[0][3]) / 1e3, float(r[0][5]) / 1e3, float(r[0][6]) / 1e3) if r else None


# ---- bench lines ------------------------------------------------------------------------------------------------------------------------
shutil.copy(f"{G}/r06_bench.json", f"{P}/r06_bench.json")
for w in ("cfg3", "cfg4", "cfg5"):
    shutil.copy(f"{G}/r06_bench_{w}.json", f"{P}/r06_bench_{w}.json")
shutil.copy(f"{G}/r06_bench_two_launches.json", f"{P}/r06_bench_two_launches.json")
if os.path.exists(f"{G}/r06_top_bench.json"):
    shutil.copy(f"{G}/r06_top_bench.json", f"{P}/r06_bench_reference_default_act.json")
if os.path.exists(f"{G}/rccl_world1.json"):
    shutil.copy(f"{G}/rccl_world1.json", f"{P}/r06_rccl_world1.json")

# ---- kernel stats -----------------------------------------------------------------------------------------------------------------------
rows = stats(f"{G}/r06_stats_kernel_stats.csv")
csv.writer(open(f"{P}/r06_kernel_stats.csv", "w")).writerows(rows)
rows2 = stats(f"{G}/r06_stats_two_kernel_stats.csv")
csv.writer(open(f"{P}/r06_kernel_stats_two_launches.csv", "w")).writerows(rows2)
rows5 = stats(f"{G}/r06_stats_cfg5_kernel_stats.csv")
csv.writer(open(f"{P}/r06_kernel_stats_cfg5.csv", "w")).writerows(rows5)

# ---- HBM traffic of the one-launch step -------------------------------------------------------------------------------------------------------
w, f = open(f"{G}/r06_write.txt").read(), open(f"{G}/r06_fetch.txt").read()
W = float(re.search(r"rcw_fill256_cast_kernel\s+WRITE_SIZE.*?mean=\s*([\d.]+)", w).group(1))
F = float(re.search(r"rcw_fill256_cast_kernel\s+FETCH_SIZE.*?mean=\s*([\d.]+)", f).group(1))
alg = 4096 * 256 * 256 * 4 / 1024
slots = 4096 * 256 * 10 / 1024
sq = open(f"{G}/r06_sq.txt").read()
open(f"{P}/r06_pmc_summary.txt", "w").write(
    "rocprofv3 PMC passes (separate runs, --pmc with --kernel-trace only), cfg2: 4096 agents x 256 columns, H_cam 256, the one-launch step\n"
    "command: rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --traffic off\n"
    "Units: KiB per dispatch (sum over the 8 XCDs).  WRITE_SIZE is exact for 16-byte streaming stores; FETCH_SIZE on gfx950\n"
    "reports 1/2 of wide streaming reads and is uncalibrated for the fill's narrow gathers (actions, slot words) and the casting half's table reads.\n"
    "(rcw_cast_successors_kernel + rcw_fill256_kernel, one dispatch each: the reset behind rcw_create, which primes the slots.)\n\n" + w + f +
    f"\nrcw_fill256_cast_kernel: algorithmic {alg:,.0f} KiB of frames per launch; WRITE_SIZE {W:,.1f} KiB (+{(W / alg - 1) * 100:.2f} %: the frames, the\n"
    f"five slots of packed column words the casting half leaves for the next launch — 10 B a column = {slots:,.0f} KiB — and the agents' state);\n"
    f"FETCH_SIZE raw {F:,.1f} KiB (x 2 = {2 * F:,.1f} KiB upper bound: actions, slot words, tile maps, state, the ray table's two rows a heading)\n\n"
    "SQ counters of the same command (per dispatch, summed over the XCDs; *_CYCLES in quad-cycles):\n" + sq)
json.dump({"workload": "cfg2", "batch": 4096, "kernel": "rcw_fill256_cast_kernel", "write_bytes_per_launch": int(W * 1024),
           "fetch_bytes_per_launch_raw": int(F * 1024), "traffic_bytes_per_launch": int(W * 1024) + 2 * int(F * 1024),
           "note": "traffic = WRITE_SIZE + 2 x FETCH_SIZE (gfx950 FETCH correction, upper bound for narrow gathers); the one-launch step (rcw_set_step_form: the rule)",
           "source": "profiles/r06_pmc_summary.txt"}, open(f"{P}/pmc_traffic.json", "w"), indent=1)

# ---- the step's two forms ------------------------------------------------------------------------------------------------------------------
forms = open(f"{G}/r06_forms_final.txt").read()
halves = open(f"{G}/r06_spec_halves_final.txt").read()
k1, k2 = kernel(rows, "rcw_fill256_cast_kernel"), kernel(rows2, "rcw_fill256_kernel<")
kc = kernel(rows2, "rcw_cast_kernel")
k5 = kernel(rows5, "rcw_fill256_cast_kernel")
def probe(w, bit):
    return float(re.search(rf"^{w} RCW_SPEC_DEBUG={bit}\s+([\d.]+) us/step", halves, re.M).group(1))
alone = " / ".join(f"{probe(w, 2):.0f}" for w in ("cfg2", "cfg3", "cfg5"))
inside = " / ".join(f"{probe(w, 0) - probe(w, 1):.0f}" for w in ("cfg2", "cfg3", "cfg5"))
memops = " / ".join(f"{probe(w, 0) - probe(w, 12):.0f}" for w in ("cfg2", "cfg3", "cfg5"))
fans = " / ".join(f"{probe(w, 0) - probe(w, 16):.0f}" for w in ("cfg2", "cfg3", "cfg5"))
cast = " / ".join(re.search(rf"^{w} two launches.*?cast\s+([\d.]+) us", halves, re.M).group(1).split(".")[0] for w in ("cfg2", "cfg3", "cfg5"))
earlier = open(f"{P}/r06_step_forms_development.txt").read() if os.path.exists(f"{P}/r06_step_forms_development.txt") else ""
open(f"{P}/r06_step_forms.txt", "w").write(
    "Round 6 — RCW.act!(env, a) SR:333-340 in ONE launch (rcw_fill256_cast_kernel) against the cast kernel followed by the fill kernel, 1 MI355X,\n"
    "the round's FINAL binary.  All figures of a block come from ONE gpurun call (boxes differ by 1-2 % in what their HBM gives a fill).\n\n"
    "== (1) bench.py, 200 steps, every BASELINE workload in both forms (tools/r06_step_forms.sh; `launch` = HIP events around the dominant launch,\n"
    "       `cast` = events around what runs in front of it, `whole step` = bytes of frames / (events around the 200 steps) against 8 TB/s)\n" + forms +
    "\n== (2) rocprofv3 --kernel-trace --stats of the headline command (profiles/r06_kernel_stats.csv, r06_kernel_stats_two_launches.csv, r06_kernel_stats_cfg5.csv)\n"
    f"   one launch:   rcw_fill256_cast_kernel  {k1[0]} calls, avg {k1[1]:.2f} us (min {k1[2]:.2f}, max {k1[3]:.2f}) = {1073741824 / k1[1] / 1e6 / 8 * 100:.1f} % of 8 TB/s on the frames' bytes\n"
    f"   two launches: rcw_fill256_kernel       {k2[0]} calls, avg {k2[1]:.2f} us (min {k2[2]:.2f}, max {k2[3]:.2f}) = {1073741824 / k2[1] / 1e6 / 8 * 100:.1f} %  +  rcw_cast_kernel avg {kc[1]:.2f} us (min {kc[2]:.2f})\n"
    f"   cfg-5, one launch: rcw_fill256_cast_kernel {k5[0]} calls, avg {k5[1]:.2f} us (min {k5[2]:.2f}, max {k5[3]:.2f}) = {8589934592 / k5[1] / 1e6 / 8 * 100:.1f} % of 8 TB/s\n"
    "\n== (3) what the casting half costs the launch: timing probes (development build, tools/r06_spec_halves.sh; RCW_SPEC_DEBUG bits: 1 = the casting\n"
    "       workgroups return at once (the fill half alone), 2 = the fill's do (the casting half alone), 4 = the casting half stores nothing,\n"
    "       8 = its turns' fans take no table loads, 16 = the current state's fan only; all but 0 give wrong frames)\n" + halves +
    f"\nReading.  The casting half alone is {alone} us of work at cfg-2 / cfg-3 / cfg-5 (five fans an agent, two divisions a column and heading in\n"
    f"place of three table loads); inside the launch it costs {inside} us (this table: whole launch - fill half alone), against {cast} us of cast\n"
    f"kernel (+ a boundary) in the two-launch step.  Of that, its stores and its turns' table loads (bits 4 + 8 off, all five fans still marched) are\n"
    f"{memops} us and the four successors' fans (bit 16: the current state's fan only) {fans} us: at cfg-5 the cost is mostly what the casting half\n"
    "writes and reads (84 MB of slot words a launch beside 8 GiB of frames on an HBM-bound launch), at cfg-3 mostly the marching of the successors.\n"
    "Both builds are compiled with every loop on a 64-byte line of the code since (5) below; before, the development build's chunk loop lay off a line\n"
    "and its launch was 30-36 us longer than the shipped one's (1271 / 1297 us at cfg-3 / cfg-5).\n" + earlier)

# ---- the top view, what it adds to a step now that the step beside it is one launch ------------------------------------------------------------
if os.path.exists(f"{G}/r06_top_shapes_plain.txt"):
    shapes = open(f"{G}/r06_top_shapes_plain.txt").read()
    hcam = open(f"{G}/r06_hcam_steps.txt").read() if os.path.exists(f"{G}/r06_hcam_steps.txt") else ""
    trows = stats(f"{G}/r06_top_kernel_stats.csv")
    csv.writer(open(f"{P}/r06_top_view_kernel_stats.csv", "w")).writerows(trows)
    open(f"{P}/r06_top_view_shapes.txt", "w").write(
        "update_top_view! (SR:446-483) over map / pixel-scale shapes (tools/top_view_shapes.py WITHOUT the profiler, 240 steps), 1 MI355X, round 6, the FINAL binary.\n"
        "The top view's kernels are round 5's in their instructions (profiles/r05_top_view_shapes.txt holds their per-kernel table (a)); the draw kernels' translation\n"
        "unit is compiled with -falign-loops=64 since this round (csrc/Makefile): rcw_fill256_draw_kernel is the camera fill's chunk loop beside the drawing in one\n"
        "function, and with that loop on a 64-byte line of the code what the top view adds to a step fell from 261 / 245 / 223 / 211 us to 219 / 222 / 214 / 206 us on\n"
        "the 80^2 / 96^2 / 104^2 / 128^2 px images (the other shapes: within a box's 1-2 %).  What else changed is the step they are compared WITH: `without the top view` is now the\n"
        "one-launch step (12-14 us shorter at these batches), so `it adds` — the whole step with the top view minus the whole step without one, the measure that\n"
        "does not depend on who hides whom — grows by that much where the handle with a top view still pays its cast kernel (every shape: the drawing needs\n"
        "the state the same launch would commit, include/rcw.h RCW_STEP_ONE_LAUNCH).  Lines as in r05 (b)/(c).\n\n" + shapes +
        "\n== the camera heights (tools/hcam_bench.py, 200 steps, the fill kernel by HIP events; 256 rows: the one-launch step)\n" + hcam)
b = line(f"{P}/r06_bench.json")
print("r06 value", round(b["value"]), "launch frac", round(b["roofline"]["frac"], 4), "step frac", round(b["roofline"]["whole_step"]["frac"], 4), "traffic", b["roofline"]["traffic"])
