#!/usr/bin/env python3
"""Dev tool (GPU box): a long rollout (default 50,000 steps x 256 agents, auto-reset on) compared with
the oracle at the end — state, episode counters, descriptors and frames.

    python tools/soak.py [steps] [top]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import raycastworlds_jl_amd as RCW
from helpers import assert_state_equal
from oracle import oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
top = len(sys.argv) > 2 and sys.argv[2] == "top"
one = len(sys.argv) > 2 and sys.argv[2] == "one"           # the one-launch step (256 agents are below where the rule takes it by itself)
B = 256
kw = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=99, auto_reset=True, out_of_bounds=1, render_top_view=top, **kw)
if top:
    env.set_top_view_form("two-kernels")                   # (256 agents would take the one-kernel form): side stream fork / join every step
if one:
    env.set_step_form("one-launch")
orc = O.OracleBatch(B, seed=99, auto_reset=1, out_of_bounds=1, render=False, **kw)
O.set_num_threads(16)
rng = np.random.default_rng(0)
t0 = time.time()
for s0 in range(0, steps, 1000):
    acts = rng.integers(1, 5, (1000, B)).astype(np.uint8)
    for a in acts:
        RCW.act_(env, a)
        orc.step(a)
    print(f"step {s0 + 1000} episodes {int(env.world.episode.sum())} ({time.time() - t0:.0f} s)", flush=True)
np.testing.assert_array_equal(env.world.episode, orc.episode)
if top:                                                    # both images of the final state against a rendering oracle given that state
    small = O.OracleBatch(B, seed=0, render_top_view=1, **kw)
    small.set_state(orc.goal, orc.position, orc.direction)
    np.testing.assert_array_equal(env.top_view_host(), small.top_view)
    np.testing.assert_array_equal(env.camera_view_host(), small.camera_view)
    print(f"top view ({env.top_view_form()}) and camera view of the final state bit-exact")
assert_state_equal(env, orc, frames=False, rays=True, where=f"after {steps} steps")
print(f"soak ok: {steps} steps x {B} agents, {int(env.world.episode.sum()) - B} episodes restarted, state and rays bit-exact")
