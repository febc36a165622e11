#!/usr/bin/env python3
"""One-off probe (GPU box): does `tensor.record_stream(<the engine's stream>)` work?  Run ONCE per variant, each in its
own process (a failure here is an abort of the process, which must not take a test run with it):

  torch   the engine runs on a stream torch owns (torch.cuda.Stream(), handed over with rcw_set_stream): record_stream on
          that Stream object — the standard idiom
  extern  the engine runs on the library's own stream, wrapped per call in torch.cuda.ExternalStream: what round 2 tried

Each: 200 steps with a temporary action tensor made on another stream and dropped right after act!, then close(), then a
few allocations (the caching allocator recycles the blocks), a final synchronisation, "probe ok".
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raycastworlds_jl_amd as RCW

variant = sys.argv[1]
print("torch", torch.__version__, "hip", torch.version.hip, "variant", variant, flush=True)
env = RCW.SingleRoomModule.SingleRoom(batch=256, seed=1, out_of_bounds=1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
if variant == "torch":
    st = torch.cuda.Stream()
    env.set_stream(st.cuda_stream)
else:
    env.set_stream(None)                                    # the library's own stream
    st = torch.cuda.ExternalStream(env.stream_ptr(), device="cuda:0")
side = torch.cuda.Stream()
for s in range(200):
    with torch.cuda.stream(side):
        a = torch.randint(1, 5, (256,), dtype=torch.uint8, device="cuda")
    st.wait_stream(side)
    env._check(env._lib.rcw_step_device(env._h, __import__("ctypes").c_void_p(a.data_ptr())))
    a.record_stream(st)
    del a
env.sync()
print("200 steps done", flush=True)
frames = env.camera_view_host(0, 2)
env.close()
print("closed", flush=True)
for _ in range(50):
    x = torch.empty(256, dtype=torch.uint8, device="cuda"); del x
torch.cuda.synchronize()
torch.cuda.empty_cache()
print("probe ok", variant, int(frames.sum() % 1000), flush=True)
