#!/usr/bin/env python3
"""Dev tool (GPU box): step latency at small batches without / with the top view, in its two-kernel and one-kernel forms
(what rcw_create's rule for the two-kernel form's smallest batch is based on).  With RCW_LIBRARY = the development build and
RCW_TOP_FUSED=0 the two-kernel form goes over the side stream again instead of the fused fill + draw launch."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import raycastworlds_jl_amd as RCW
CFG2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
for B in (1, 16, 64, 256, 1024):
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
