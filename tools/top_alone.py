#!/usr/bin/env python3
"""Dev tool (GPU box): rcw_update_top_view ALONE (no camera fill beside the drawing) in a given form, for tools/kprof.sh / rocprofv3 --pmc.

    python tools/top_alone.py H,W,pu,N one-kernel|two-kernels|auto [reps=40] [batch]

two-kernels needs the development library for geometries the shipped rule sends elsewhere (RCW_LIBRARY=...librcw_hip_dev.so,
RCW_TOP_ALONE_SPLIT=1 is set here); prints the HIP-event time of a call.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
form = sys.argv[2] if len(sys.argv) > 2 else "auto"
if form == "two-kernels":
    os.environ.setdefault("RCW_TOP_ALONE_SPLIT", "1")
import raycastworlds_jl_amd as RCW

H, W, pu, N = (int(v) for v in sys.argv[1].split(","))
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
px = H * pu * W * pu
B = int(sys.argv[4]) if len(sys.argv) > 4 else max(64, min(65536, (1 << 30) // (4 * px)))
env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, out_of_bounds=1, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N,
                                      pu_per_tu=pu, render_top_view=True, library=os.environ.get("RCW_LIBRARY"))
if form != "auto":
    env.set_top_view_form(form)
for _ in range(5):
    RCW.update_top_view_(env)
env.sync(); env.timer_start()
for _ in range(reps):
    RCW.update_top_view_(env)
ms = env.timer_stop() / reps
gib = 4 * px * B / 2 ** 30
print(f"alone map {H:2d}x{W:2d} pu {pu:2d} N {N:4d} image {H * pu:4d}x{W * pu:4d} B {B:5d} asked {form:11s} step-form {env.top_view_form():11s}: "
      f"{ms * 1e3:7.1f} us a call = {ms * 1e3 / gib:6.1f} us / GiB = {4 * px * B / ms / 1e6 / 80:4.1f} % of 8 TB/s", flush=True)
env.sync(); env.close()
