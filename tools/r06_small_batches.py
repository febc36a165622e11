#!/usr/bin/env python3
"""Dev tool (GPU box): a step's two forms at small batches, where a step is launch-bound: us per step of direct calls (one-launch step,
two-launch step) and of a captured graph of the two-launch step (a captured step takes that form), cfg-2's geometry.

    python tools/r06_small_batches.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycastworlds_jl_amd as RCW

GEOMETRIES = {"cfg2": dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256), "cfg3": dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=512),
              "cfg5": dict(height_tile_map_tu=32, width_tile_map_tu=32, num_rays=1024), "cfg1": dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)}
cases = [("cfg2", b) for b in (1, 4, 16, 64, 256, 1024, 4096)]
if len(sys.argv) > 1:                                     # geometry:B ...
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for geo, B in cases:
    CFG2 = GEOMETRIES[geo]
    row = {}
    for form in ("one-launch", "two-launches"):
        env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, **CFG2)
        env.set_step_form(form)
        stream = torch.cuda.Stream()
        env.set_stream(stream.cuda_stream)
        torch.cuda.set_stream(stream)
        actions = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
        steps = 3000 if B <= 256 else 400

        def loop(fn):
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e6

        row[form] = loop(lambda: RCW.act_(env, actions))
        if form == "two-launches":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(8):
                    RCW.act_(env, actions)
            row["graph of 8 two-launch steps"] = loop(g.replay) / 8
        env.sync(); env.close()
    print(f"{geo} B={B:5d} ({B * CFG2['num_rays'] // 1024:5d} MiB of frames): " + " | ".join(f"{k} {v:7.2f} us/step" for k, v in row.items()), flush=True)
