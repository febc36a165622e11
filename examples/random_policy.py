#!/usr/bin/env python3
"""The README walkthrough of RayCastWorlds.jl (README.md:15-67) for a batch of agents on one MI355X.

    python examples/random_policy.py [batch] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raycastworlds_jl_amd as RCW

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200

env = RCW.SingleRoomModule.SingleRoom(batch=batch, auto_reset=True, out_of_bounds=1)   # reference defaults: 8 x 16, 512 rays
RCW.reset_(env)                                   # RCW.reset!(env)
print(RCW.get_action_names(env))                  # RCW.get_action_names(env)
RCW.act_(env, 1)                                  # RCW.act!(env, 1): every agent moves forward

rl = RCW.RLBaseEnv(env)                           # RCW.RLBaseEnv(env)
RLBase = RCW.RLBase
RLBase.reset_(rl)
rng = np.random.default_rng(0)
returns = np.zeros(batch, np.float32)
t0 = time.perf_counter()
for _ in range(steps):
    state = RLBase.state(rl)                      # device-resident (B, N, H_cam) uint32, aliased
    rl(rng.integers(1, 5, batch))                 # rl(action)
    returns += RLBase.reward(rl)
env.sync()
dt = time.perf_counter() - t0
print(f"{batch} agents x {steps} random steps on {env.device_name()}: {batch * steps / dt / 1e6:.2f} M env-steps/s "
      f"(host-driven loop), goals reached: {int(returns.sum())}, state {state}")
RCW.save_agent_ppm(env, 0, "/tmp/agent0.ppm")
print("agent 0's camera view written to /tmp/agent0.ppm")
