#!/usr/bin/env python3
"""Dev tool (GPU box): is a captured HIP graph of one step (cast + fill) cheaper than the two direct launches when
the step is launch-bound (small batches)?  Uses torch's CUDAGraph capture on the stream the engine shares.

    python tools/graph_bench.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycastworlds_jl_amd as RCW

CFG2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
for B in (1, 16, 64, 256, 1024, 4096):
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=1, auto_reset=True, out_of_bounds=1, **CFG2)
    stream = torch.cuda.Stream()
    env.set_stream(stream.cuda_stream)
    torch.cuda.set_stream(stream)
    actions = torch.randint(1, 5, (B,), dtype=torch.uint8, device="cuda")
    steps = 2000 if B <= 256 else 300

    def loop(fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    direct = loop(lambda: RCW.act_(env, actions))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        RCW.act_(env, actions)
    graph1 = loop(g.replay)
    g8 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g8, stream=stream):
        for _ in range(8):
            RCW.act_(env, actions)
    graph8 = loop(g8.replay) / 8
    print(f"B={B:5d}: direct {direct:7.2f} us/step | graph of 1 step {graph1:7.2f} | graph of 8 steps {graph8:7.2f} us/step", flush=True)
    env.sync()
    env.close()
