import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import raycastworlds_jl_amd as RCW
for B in (1, 64, 1024):
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    st = torch.cuda.Stream(); env.set_stream(st); torch.cuda.set_stream(st)
    a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
    ah = a.cpu().numpy()
    for name, act in (("device tensor", a), ("host array", ah), ("scalar", 3)):
        for _ in range(200): RCW.act_(env, act)
        env.sync(); t = time.perf_counter()
        n = 5000
        for _ in range(n): RCW.act_(env, act)
        t_issue = time.perf_counter() - t
        env.sync(); t_all = time.perf_counter() - t
        print(f"B {B:5d} {name:14s}: {t_issue / n * 1e6:6.1f} us of host time a call, {t_all / n * 1e6:6.1f} us a step end to end")
    env.close()
