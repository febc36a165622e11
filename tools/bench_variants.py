#!/usr/bin/env python3
"""Dev tool (GPU box): step time of non-default variants (Float64 world units, top view)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycastworlds_jl_amd as RCW

CFG2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
for name, kw in (("Float32", {}), ("Float64", dict(T="Float64")), ("Float32 + top view", dict(render_top_view=True)),
                 ("Float64 32x32 N=1024 B=2048", dict(T="Float64", height_tile_map_tu=32, width_tile_map_tu=32, num_rays=1024))):
    B = 2048 if "32x32" in name else 4096
    cfg = {**CFG2, **kw}
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=0, out_of_bounds=1, auto_reset=True, **cfg)
    a = torch.randint(1, 5, (120, B), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for s in range(20):
        RCW.act_(env, a[s])
    env.sync()
    env.profile(True)
    t0 = time.perf_counter()
    for s in range(20, 120):
        RCW.act_(env, a[s])
    env.sync()
    dt = (time.perf_counter() - t0) / 100
    c, _t, f, n = env.profile_read()
    print(f"{name:32s} {dt * 1e6:8.1f} us/step  cast(+top view) {c * 1e3:7.1f} us  fill {f * 1e3:7.1f} us  {B / dt / 1e6:6.2f} M env-steps/s")
    env.close()
