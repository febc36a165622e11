#!/usr/bin/env python3
"""Assemble profiles/<tag>_top_view_shapes.txt from what tools/gpu_round.sh left under gpurun_out/ (<tag>_top_shapes_kernels.txt:
rocprofv3 kernel averages per shape; <tag>_top_shapes_steps.txt: the HIP-event lines of tools/top_view_shapes.py).
usage: python tools/top_shapes_profile.py r03"""
import re
import sys

tag = sys.argv[1]
out = [f"""update_top_view! (SR:446-483) over map / pixel-scale shapes (tools/top_view_shapes.py), 1 MI355X, round {int(tag[1:])}.  ~1 GiB of top view per
launch (65,536 agents at most; round 3 capped the batch at 16,384, i.e. 0.4-0.7 GiB for images of 80-104 px).  Two measurements of the same runs:
  (a) rocprofv3 --kernel-trace --stats per shape (tools/kprof.sh): the kernels' own average / minimum / median durations over 240 steps (a run's first ten to twenty
      launches are 5-15 % slower than the rest — clocks —: the median says what a launch takes once they are up),
      and the store kernel's bandwidth on the algorithmic bytes 4*(H*pu)*(W*pu)*B against the 8 TB/s HBM peak;
  (b) HIP events of the step (start | behind the cast kernel | behind the camera fill | end): "in a step" = from the camera fill's end to the
      step's end (the store kernel incl. any wait for the draw kernel — what the draw kernel does not hide behind the camera fill shows
      up here), the camera fill beside it (from the cast kernel's end to the fill's end), and rcw_update_top_view alone with the form it takes
      (round 5: draw -> store back to back from 256 x 256 px, at pixel scales that are no multiple of 4 and below 16 px a tile; else the
      one-kernel form), also in us per GiB of top view.  Round 5 also sends the camera fill to the side stream where it is the shorter of
      the two (768^2, 1024^2 px below): the drawing then hides the FILL, "in a step" shrinks to the store kernel and the fill's column
      grows — so the line ends with the measure that does not depend on who hides whom: the whole step (start -> end) with the top view,
      the whole step of the same geometry WITHOUT one (cast + fill), and their difference = what the top view ADDS to a step, as a
      share of the HBM peak on the top view's bytes.  (Where the camera fill is 0.6-1.6 ms — the four shapes of 80^2 ... 128^2 px images —
      that difference of two whole steps moves by +-25 us from run to run, 1.5 % of a step: two collections of this table gave 175 and 214 us
      at 80^2 px, 172 / 193 at 96^2, 173 / 190 at 104^2, 177 / 184 at 128^2; "in a step" is the steadier number there: 176-181 us.)
Kernels: rcw_top_store_kernel (whole 256-row chunks: pu in {{8..256}} dividing 256, H*pu % 256 == 0), rcw_top_store_flat_kernel
<STRADDLE, NARROW, K> (any pu >= 9, H*pu % 4 == 0 — 256-pixel chunks of the flat batch, K columns a chunk), rcw_top_draw_kernel.

== (a) per-kernel, rocprofv3 (us: average, minimum, median)"""]
for line in open(f"gpurun_out/{tag}_top_shapes_kernels.txt"):
    m = re.match(r"top_(\d+),(\d+),(\d+),(\d+)\s+(rcw_\S+(?: \S+)*?)\s+calls\s+\d+ avg\s+([\d.]+) us\s+min\s+([\d.]+)(?:\s+p50\s+([\d.]+))?", line)
    if not m:
        continue
    H, W, pu, N = (int(m.group(k)) for k in range(1, 5))
    name, avg, mn = m.group(5), float(m.group(6)), float(m.group(7))
    p50 = float(m.group(8)) if m.group(8) else None
    px = H * pu * W * pu
    B = max(64, min(65536, (1 << 30) // (4 * px)))
    row = f"map {H:2d}x{W:2d} pu {pu:2d} N {N:4d}  {name:46s} avg {avg:7.1f}  min {mn:7.1f}" + (f"  p50 {p50:7.1f}" if p50 else "")
    if "store" in name:
        by = 4 * px * B
        row += (f"    {by / avg / 1e6:4.2f} TB/s = {by / avg / 1e6 / 8 * 100:4.1f} %" + (f" (median launch {by / p50 / 1e6 / 8 * 100:4.1f} %," if p50 else " (")
                + f" best launch {by / mn / 1e6 / 8 * 100:4.1f} %)")
    out.append(row)
out.append("\n== (b) inside a step, HIP events (240 steps)")
out += [l.rstrip("\n") for l in open(f"gpurun_out/{tag}_top_shapes_steps.txt")]
import os
if os.path.exists(f"gpurun_out/{tag}_top_shapes_plain.txt"):
    out.append("\n== (c) the same tool run WITHOUT rocprofv3 (its tracing moves cross-stream timings by a few us; 240 steps): the line's last part is the whole-step\n"
               "difference WITHOUT the per-kernel events as well — two events around all the steps, with and without the top view: the cleanest figure of what the top view adds")
    out += [l.rstrip("\n") for l in open(f"gpurun_out/{tag}_top_shapes_plain.txt")]
open(f"profiles/{tag}_top_view_shapes.txt", "w").write("\n".join(out) + "\n")
print(f"profiles/{tag}_top_view_shapes.txt: {len(out)} lines")
