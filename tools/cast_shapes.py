#!/usr/bin/env python3
"""Dev tool (GPU box): cast / fill kernel times for unusual shapes (columns, batch, map size, world-unit type)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycastworlds_jl_amd as RCW

for H, W, N, B, T in ((8, 8, 256, 4096, "Float32"), (8, 8, 64, 16384, "Float32"), (8, 8, 33, 16384, "Float32"),
                      (8, 8, 4096, 256, "Float32"), (8, 8, 1, 65536, "Float32"), (64, 64, 256, 4096, "Float32"),
                      (200, 200, 256, 1024, "Float32"), (8, 8, 256, 4096, "Float64"), (32, 32, 1024, 8192, "Float64")):
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, height_tile_map_tu=H,
                                          width_tile_map_tu=W, num_rays=N, T=T)
    st = torch.cuda.Stream(); env.set_stream(st.cuda_stream); torch.cuda.set_stream(st)
    a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        RCW.act_(env, a)
    env.profile(True)
    for _ in range(30):
        RCW.act_(env, a)
    c, t, f, n = env.profile_read(); env.profile(False)
    by = 4 * 256 * N * B
    print(f"map {H:3d}x{W:3d} N {N:5d} B {B:6d} {T}: cast {c * 1e3:8.1f} us  fill {f * 1e3:8.1f} us ({by / f / 1e6 / 80:4.1f} %)  "
          f"cast share {c / (c + f) * 100:4.1f} %", flush=True)
    try:
        env.sync()
    except IndexError:
        env.clear_error()
    env.close()
