#!/usr/bin/env python3
"""Dev tool (GPU box): handle churn / leak check, a 65536-agent batch, and the PCIe-inclusive rate
when the caller insists on host-resident observations (never bench.py's `value`)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raycastworlds_jl_amd as RCW

CFG2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
env = RCW.SingleRoomModule.SingleRoom(batch=512, seed=0, **CFG2); RCW.act_(env, 1); env.sync(); env.close()   # one-time context cost
free0, _ = torch.cuda.mem_get_info()
for k in range(40):
    env = RCW.SingleRoomModule.SingleRoom(batch=512, seed=k, **CFG2)
    RCW.act_(env, 1)
    env.sync()
    env.close()
free1, _ = torch.cuda.mem_get_info()
print(f"handle churn: 40 create/step/destroy cycles, free memory delta {(free0 - free1) / 2**20:.1f} MiB")

env = RCW.SingleRoomModule.SingleRoom(batch=65536, seed=1, out_of_bounds=1,
                                      height_tile_map_tu=16, width_tile_map_tu=16, num_rays=256)
a = torch.randint(1, 5, (20, 65536), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for s in range(4):
    RCW.act_(env, a[s])
env.sync()
t0 = time.perf_counter()
for s in range(4, 20):
    RCW.act_(env, a[s])
env.sync()
dt = (time.perf_counter() - t0) / 16
print(f"cfg4 on ONE GPU (65536 agents, 16 GiB of frames): {dt * 1e3:.2f} ms/step, {65536 / dt / 1e6:.1f} M env-steps/s, "
      f"{65536 * 262144 / dt / 1e12:.2f} TB/s")
env.close()

env = RCW.SingleRoomModule.SingleRoom(batch=4096, seed=1, out_of_bounds=1, **CFG2)
acts = np.random.default_rng(0).integers(1, 5, (6, 4096)).astype(np.uint8)
host = np.empty((4096, 256, 256), dtype=np.uint32)
import ctypes as C
from raycastworlds_jl_amd import _capi
RCW.act_(env, acts[0]); env.sync()
t0 = time.perf_counter()
for s in range(1, 6):
    RCW.act_(env, acts[s])                                                   # host actions (4 KiB H2D)
    _capi.check(env._lib.rcw_obs_copy(env._h, host.ctypes.data_as(C.c_void_p), 0, 4096))   # 1 GiB D2H
dt = (time.perf_counter() - t0) / 5
print(f"PCIe-inclusive (host actions in, host frames out, pageable): {dt * 1e3:.1f} ms/step, "
      f"{4096 / dt / 1e6:.3f} M env-steps/s, {2**30 / dt / 1e9:.1f} GB/s D2H")
env.close()
