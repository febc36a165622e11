#!/usr/bin/env python3
"""A device-resident acting loop: the engine runs on torch's current stream, observations never leave
HBM, a stand-in "policy" (a strided mean over the frame + argmax) picks the next actions with torch ops.

    python examples/torch_loop.py [batch] [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycastworlds_jl_amd as RCW

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200

env = RCW.SingleRoomModule.SingleRoom(batch=batch, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256,
                                      auto_reset=True, out_of_bounds=1)
env.set_stream(torch.cuda.current_stream().cuda_stream)      # engine and torch share one stream: no host syncs
obs = env.camera_view.torch().view(torch.int32)              # (B, N, H_cam), aliases the engine's buffer
reward = env.reward_device().torch()
done = env.done_device().torch().view(torch.uint8)
w = torch.randn(4, 16, device="cuda")

episodes = torch.zeros((), device="cuda")
returns = torch.zeros((), device="cuda")
t0 = 0.0
for it in range(steps + 10):
    if it == 10:                                             # the first iterations load torch's BLAS kernels
        torch.cuda.synchronize()
        episodes.zero_(); returns.zero_()
        t0 = time.perf_counter()
    feat = (obs[:, ::16, 128] & 0xFF).float()                # 16 grey samples along the horizon row
    logits = feat @ w.T + torch.randn(batch, 4, device="cuda")
    actions = (logits.argmax(dim=1) + 1).to(torch.uint8)     # 1..4
    RCW.act_(env, actions)
    returns += reward.sum()
    episodes += done.sum()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{batch} agents x {steps} steps, policy + engine on one stream: {batch * steps / dt / 1e6:.2f} M env-steps/s, "
      f"{int(episodes)} episodes ended, return {float(returns):.0f}")
