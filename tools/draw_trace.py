"""Dev tool (GPU box): where a workgroup of the top view's draw kernel spends its life.  Needs the measurement build WITH the
development switches (`make -C raycastworlds.jl_amd/csrc trace EXTRA=-DRCW_DEV_SWITCHES` -> lib/librcw_hip_trace.so, loaded here by
path): the first wavefront of the draw workgroups of agents 0..2047 leaves s_memrealtime (100 MHz) at entry, behind the cleared planes'
barrier, when its rays are cast and the lines set up, when the lines are walked, behind the second barrier, and when the planes are
copied out and acknowledged.  rcw_update_top_view alone, two-kernel form (draw -> store back to back): nothing runs beside the drawing.

    python tools/draw_trace.py H,W,pu,N [reps=6] [batch]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RCW_TOP_ALONE_SPLIT", "1")
import numpy as np

import raycastworlds_jl_amd as RCW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RCW_LIBRARY") or os.path.join(ROOT, "raycastworlds.jl_amd", "lib", "librcw_hip_trace.so")
H, W, pu, N = (int(v) for v in sys.argv[1].split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
px = H * pu * W * pu
B = int(sys.argv[3]) if len(sys.argv) > 3 else max(64, min(65536, (1 << 30) // (4 * px)))
env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, out_of_bounds=1, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N,
                                      pu_per_tu=pu, render_top_view=True, library=LIB)
env.set_top_view_form("two-kernels")
lib = env._lib
lib.rcw_draw_trace_read.argtypes = [ctypes.c_void_p]
buf = np.zeros(2048 * 20, dtype=np.uint64)
n = min(B, 2048)
names = ["planes cleared, barrier", "rays cast, lines set up", "lines walked", "barrier", "planes copied out"]
print(f"rcw_top_draw_kernel, map {H}x{W} at {pu} px a tile ({H * pu}x{W * pu} px), {N} rays, {B} agents; first wavefront of the first {n} workgroups; us")
print(" launch | call (draw + store) | first entry -> last end of these | wave life median / p95 | longest line of wave 0 (steps) | " + " | ".join(names))
for rep in range(reps):
    env.sync(); env.timer_start()
    RCW.update_top_view_(env)
    call = env.timer_stop() * 1e3
    assert lib.rcw_draw_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(2048, 20)[:n].astype(np.int64)
    st = t[:, :6]
    t0 = st[:, 0].min()
    life = (st[:, 5] - st[:, 0]) / 100.0
    seg = np.diff(st, axis=1) / 100.0
    span = (st[:, 5].max() - t0) / 100.0
    ent = np.sort((st[:, 0] - t0) / 100.0); hw = t[:, 6]
    cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 8) & 0xF) + 16 * ((hw >> 12) & 0x1) + 32 * ((hw >> 13) & 0x7)      # XCC, CU, SH, SE of HW_ID
    if rep == reps - 1:
        # how many of the traced workgroups are alive at once (entry .. end), at its peak and on average over the span
        ev = np.concatenate([np.stack([st[:, 0], np.ones(n, np.int64)], 1), np.stack([st[:, 5], -np.ones(n, np.int64)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        alive = np.cumsum(ev[:, 1])
        print(f"   entries of the {n} traced workgroups: 10 % by {ent[n // 10]:.1f} us, 50 % by {ent[n // 2]:.1f}, 90 % by {ent[9 * n // 10]:.1f}, last {ent[-1]:.1f}; alive at once: peak {alive.max()}, "
              f"mean {((st[:, 5] - st[:, 0]).sum() / max(1, st[:, 5].max() - t0)):.0f}; distinct (XCC, SE, SH, CU) seen: {len(set(cu.tolist()))}")
    print(f" {rep:6d} | {call:19.1f} | {span:32.2f} | {np.median(life):9.2f} / {np.percentile(life, 95):5.2f}   | {np.median(t[:, 8] & 0xFFFFFFFF):29.0f} | " +
          " | ".join(f"{np.median(seg[:, k]):{len(names[k])}.2f}" for k in range(5)))
env.sync(); env.close()
