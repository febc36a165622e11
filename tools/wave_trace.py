"""Dev tool (GPU box): what rcw_fill256_kernel's 1024 wavefronts do in time.  Needs the measurement build of the library
(`make -C raycastworlds.jl_amd/csrc trace` -> lib/librcw_hip_trace.so, loaded here by path; the package never loads it):
every wavefront leaves the s_memrealtime (100 MHz, one clock for the device) at which each of its 16 groups starts its
descriptor loads and has them back, when it ends, and where it ran (HW_ID, XCC_ID).

Prints, for cfg-2 (1 GiB a launch): the kernel's length, the pause a group's prefetch makes, how far apart the wavefronts
run, and the end time by XCD / shader engine.  usage: python tools/wave_trace.py [reps]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raycastworlds_jl_amd as RCW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RCW_LIBRARY") or os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "librcw_hip_trace.so")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
HC = int(sys.argv[2]) if len(sys.argv) > 2 and "," not in sys.argv[2] else 256   # camera height: 256 traces rcw_fill256_kernel, any flat height rcw_fill_flat_kernel
TOP = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 and "," in sys.argv[2] else None   # H,W,pu: rcw_top_store_flat_kernel
if TOP:                                                       # (the camera is kept small — 24 rows, 64 columns — so that the trace is the top view's store kernel's, which runs last)
    B = max(64, min(16384, (1 << 30) // (4 * TOP[0] * TOP[2] * TOP[1] * TOP[2])))
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, library=LIB, height_camera_view_pu=20,
                                          height_tile_map_tu=TOP[0], width_tile_map_tu=TOP[1], num_rays=64, render_top_view=True, pu_per_tu=TOP[2])
    env.set_top_view_form("two-kernels", runs=1)
else:
    B = (1 << 30) // (4 * 256 * HC)
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, library=LIB, height_camera_view_pu=HC,
                                          height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
buf = np.zeros(1024 * 40, dtype=np.uint64)
env.sync()
waves = np.arange(1024)
print((f"rcw_top_store_flat_kernel, map {TOP[0]}x{TOP[1]} at {TOP[2]} pixels a tile, {B} agents" if TOP else f"{env.fill_kernel_name()}, {B} agents x 256 columns x {HC} rows") + f" (~1 GiB a launch), {reps} launches traced; times in us")
print(" launch | kernel | prefetch pause median / max | restart spread | end spread | group period | ends: even XCDs / odd XCDs")
ends = []
for rep in range(reps):
    for _ in range(4):
        RCW.act_(env, a)
    env.sync()
    assert (env._lib.rcw_top_store_trace_read if TOP else env._lib.rcw_wave_trace_read)(buf.ctypes.data_as(ctypes.c_void_p)) == 0   # (each translation unit has its own trace array)
    t = buf.reshape(1024, 20, 2).astype(np.int64)
    ng = int(min(16, (t[:, :18, 0] > 0).sum(axis=1).min()))      # groups every wavefront traced (16 at 1 GiB; fewer for smaller batches)
    t0, t1, tend, hw = t[:, :ng, 0], t[:, :ng, 1], t[:, 19, 0], t[:, 18, 0]
    xcc = (hw >> 32) & 0xF
    start = t0[:, 0].min()
    pause = (t1 - t0)[:, 1:] / 100.0
    spread = ((t1.max(axis=0) - t1.min(axis=0)) / 100.0)[1:].mean()
    period = np.diff(np.median(t1, axis=0)).mean() / 100.0
    e = (tend - start) / 100.0
    ends.append(e)
    print(f" {rep:6d} | {e.max():6.1f} | {np.median(pause):5.2f} / {pause.max():5.2f}               | {spread:14.1f} | {e.max() - e.min():10.1f} | {period:12.2f} |"
          f" {e[xcc % 2 == 0].mean():6.1f} / {e[xcc % 2 == 1].mean():6.1f}")
ends = np.array(ends)
# the last traced launch in detail: where do the odd XCDs lose their time — in the prefetch pause or in the store loop?
loop = (t0[:, 1:] - t1[:, :-1]) / 100.0                      # 64 chunks' stores of group k (issue time, incl. back-pressure)
pz = (t1 - t0) / 100.0
for name, sel in (("even XCDs", xcc % 2 == 0), ("odd XCDs ", xcc % 2 == 1)):
    print(f"{name}: prefetch pause by group (median over wavefronts):", np.round(np.median(pz[sel], axis=0), 2).tolist())
    print(f"{name}: store loop by group:", np.round(np.median(loop[sel], axis=0), 2).tolist())
    print(f"{name}: sum of pauses {np.median(pz[sel].sum(axis=1)):.1f}, sum of store loops {np.median(loop[sel].sum(axis=1)):.1f}, first group starts at {np.median(t0[sel, 0] - start) / 100.0:.2f}")
hwid = hw & 0xFFFFFFFF
cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
print("XCC_ID of workgroups 0..15:", xcc[::4][:16].tolist(), "| wavefronts per XCC:", np.bincount(xcc, minlength=8).tolist(),
      "| distinct CUs:", len(np.unique(xcc * 4096 + se * 512 + sh * 256 + cu)))
print("mean end time by XCC_ID:", np.round([ends[:, xcc == x].mean() for x in range(8)], 1).tolist())
print("mean end time by shader engine within XCC 0 / XCC 1:", np.round([ends[:, (xcc == 0) & (se == x)].mean() for x in range(4)], 1).tolist(),
      "/", np.round([ends[:, (xcc == 1) & (se == x)].mean() for x in range(4)], 1).tolist())
last = ends[-1]
order = np.argsort(-last)[:12]
simd = (hwid >> 4) & 3
print("the 12 wavefronts that ended last (wavefront, XCC, SE, CU, SIMD: end, its pauses, its store loops):")
for w in order:
    print(f"  {int(w):5d}  xcc {int(xcc[w])} se {int(se[w])} cu {int(cu[w]):2d} simd {int(simd[w])}: {last[w]:6.1f}  {pz[w].sum():5.1f}  {loop[w].sum():6.1f}   loops by group {np.round(loop[w], 1).tolist()}")
print("end time percentiles within the odd XCDs (min, 10, 50, 90, 99, max):", np.round(np.percentile(last[xcc % 2 == 1], [0, 10, 50, 90, 99, 100]), 1).tolist())
c = np.corrcoef(ends)
print("correlation of the per-wavefront end times between launches:", round(float((c.sum() - reps) / (reps * reps - reps)), 3))
env.close()
