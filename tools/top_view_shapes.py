#!/usr/bin/env python3
"""Dev tool (GPU box): top view kernel time / bandwidth for a few map and pixel-scale shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycastworlds_jl_amd as RCW

SHAPES = ((8, 8, 32, 256), (8, 16, 32, 512), (16, 16, 32, 256), (8, 8, 16, 256), (12, 12, 32, 256), (8, 8, 13, 256),
                    (8, 8, 64, 256), (32, 32, 32, 1024), (8, 8, 32, 64), (32, 32, 8, 256), (24, 24, 32, 256), (9, 9, 32, 256), (9, 12, 32, 256),
                    (8, 8, 10, 256), (8, 8, 12, 256), (8, 8, 20, 256), (8, 8, 24, 256), (8, 16, 24, 512), (16, 16, 20, 256))
if len(sys.argv) > 1:                                    # one shape: H,W,pu,N
    SHAPES = (tuple(int(v) for v in sys.argv[1].split(",")),)
for H, W, pu, N in SHAPES:
    px = H * pu * W * pu
    B = max(64, min(65536, (1 << 30) // (4 * px)))          # ~1 GiB of top view a launch (round 3 capped the batch at 16,384 agents: 0.4-0.7 GiB for the small images)
    if os.environ.get("TOPSHAPES_BATCH"):                 # development: another batch than ~1 GiB of top view
        B = int(os.environ["TOPSHAPES_BATCH"])
    extra = {}
    if os.environ.get("TOPSHAPES_HCAM"):                  # development: another camera height (no fused fill + draw launch then: the side stream)
        extra["height_camera_view_pu"] = int(os.environ["TOPSHAPES_HCAM"])
    if os.environ.get("TOPSHAPES_RADIUS"):                # development: another player radius (the circle drawn into the top view, SR:480)
        extra["player_radius_wu"] = float(os.environ["TOPSHAPES_RADIUS"])
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, height_tile_map_tu=H,
                                          width_tile_map_tu=W, num_rays=N, pu_per_tu=pu, render_top_view=True, **extra)
    if os.environ.get("TOPSHAPES_RUNS"):                  # development: the two-kernel form in this many runs of agents
        env.set_top_view_form("two-kernels", runs=int(os.environ["TOPSHAPES_RUNS"]))
    st = torch.cuda.Stream(); env.set_stream(st.cuda_stream); torch.cuda.set_stream(st)
    a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
    STEPS = int(os.environ.get("TOPSHAPES_STEPS", "20"))
    for _ in range(max(3, STEPS // 4)):
        RCW.act_(env, a)
    env.profile(True)
    for _ in range(STEPS):
        RCW.act_(env, a)
    c, t, f, n = env.profile_read(); env.profile(False)
    env.sync(); env.timer_start()                         # ... and the same steps without the per-kernel events: two events around all of them
    for _ in range(STEPS):
        RCW.act_(env, a)
    plain = env.timer_stop() / STEPS
    by = 4 * px * B
    # the stand-alone call (two-kernel form: draw, then store, back to back on one stream)
    env.sync(); env.timer_start()
    for _ in range(10):
        RCW.update_top_view_(env)
    alone = env.timer_stop() / 10
    gib = by / 2 ** 30
    form, alone_form = env.top_view_form(), env.update_top_view_form()
    env.sync(); env.close()
    # ... and the same step WITHOUT a top view (cast + camera fill): what the top view ADDS to a step is the difference of the two
    # steps' whole times — the measure that does not depend on which kernel runs on which stream, or on what "beside" hides
    bare = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N, **extra)
    bare.set_stream(st.cuda_stream)
    for _ in range(max(3, STEPS // 4)):
        RCW.act_(bare, a)
    bare.profile(True)
    for _ in range(STEPS):
        RCW.act_(bare, a)
    c0, t0, f0, n0 = bare.profile_read(); bare.profile(False)
    bare.sync(); bare.timer_start()
    for _ in range(STEPS):
        RCW.act_(bare, a)
    plain0 = bare.timer_stop() / STEPS
    bare.sync(); bare.close()
    adds = (c + t + f) - (c0 + t0 + f0)
    print(f"map {H:2d}x{W:2d} pu {pu:2d} N {N:4d} image {H * pu:4d}x{W * pu:4d} B {B:5d} {form:11s}: in a step {t * 1e3:7.1f} us "
          f"{by / t / 1e6:6.0f} GB/s ({by / t / 1e6 / 80:4.1f} %), camera fill beside it {f * 1e3:7.1f} us, stand-alone ({alone_form}) {alone * 1e3:7.1f} us "
          f"= {alone * 1e3 / gib:6.1f} us / GiB; a step {1e3 * (c + t + f):7.1f} us, without the top view {1e3 * (c0 + t0 + f0):7.1f}: it adds {adds * 1e3:7.1f} us "
          f"= {by / adds / 1e6 / 80:4.1f} % of the HBM peak on its bytes; without the per-kernel events ({STEPS} steps between two events): "
          f"{plain * 1e3:7.1f} - {plain0 * 1e3:7.1f} = {(plain - plain0) * 1e3:7.1f} us = {by / (plain - plain0) / 1e6 / 80:4.1f} %", flush=True)
