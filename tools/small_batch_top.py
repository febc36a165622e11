#!/usr/bin/env python3
"""Dev tool (GPU box): step latency at small batches without / with the top view, in its two-kernel and one-kernel forms
(what the 256 MiB rule in rcw_create is based on)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import raycastworlds_jl_amd as RCW
CFG2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
for B in (1, 64, 256, 1024):
    for top in (False, True):
        for split in ("1", "0") if top else ("1",):
            env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, render_top_view=top, **CFG2)
            if top:
                env.set_top_view_form("two-kernels" if split == "1" else "one-kernel")
            st = torch.cuda.Stream(); env.set_stream(st.cuda_stream); torch.cuda.set_stream(st)
            a = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
            for _ in range(50): RCW.act_(env, a)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(1000): RCW.act_(env, a)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 1000 * 1e6
            print(f"B={B:5d} top={top} form={env.top_view_form():11s}: {dt:7.1f} us/step", flush=True)
            env.sync(); env.close()
