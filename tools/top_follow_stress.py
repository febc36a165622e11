#!/usr/bin/env python3
"""Dev tool (GPU box): two render_top_view handles with big planes (1024 x 1024 and 768 x 768 px, batch 3) stepped alternately in the
automatic and the two-kernel form, every pixel against the oracle after every step — the stress under which the development
experiment RCW_TOP_FOLLOW (the store kernel following the draw kernel, docs/experiments.md) showed its first protocol bug.

    RCW_LIBRARY=.../librcw_hip_dev.so RCW_TOP_FOLLOW=3 python tools/top_follow_stress.py big|small|both
"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import raycastworlds_jl_amd as RCW
from oracle import oracle
big = dict(pu_per_tu=32, height_tile_map_tu=32, width_tile_map_tu=32, num_rays=128)
small = dict(pu_per_tu=32, height_tile_map_tu=24, width_tile_map_tu=24, num_rays=128)
def make(B, seed, **kw):
    env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=seed, render_top_view=1, out_of_bounds=1, **kw)
    orc = oracle.OracleBatch(B, seed=seed, render_top_view=1, out_of_bounds=1, **kw)
    return env, orc
which = sys.argv[1]
rng = np.random.default_rng(4)
envs = []
if which in ("big", "both"): envs.append(make(3, 5, **big))
if which in ("small", "both"): envs.append(make(3, 6, **small))
for form in (None, "two-kernels"):
    for e, o in envs: e.set_top_view_form(form)
    for s in range(4):
        a = rng.integers(1, 5, 3).astype(np.uint8)
        import time
        for e, o in envs:
            t0 = time.perf_counter()
            RCW.act_(e, a)
            try:
                e.sync()
            except Exception as ex:
                print("  sync:", str(ex)[:150])
            print("  step took %.1f ms" % ((time.perf_counter() - t0) * 1e3))
            o.step(a)
        for i, (e, o) in enumerate(envs):
            got = e.top_view_host(); bad = int((got != o.top_view).sum())
            print(which, "form", form, "step", s, "env", i, e.top_view_form(), e.update_top_view_form(), "mismatching pixels", bad, flush=True)
            if bad:
                d = np.argwhere(got != o.top_view)
                print("  first", d[:3].tolist(), "agents", sorted(set(d[:, 0].tolist())) if d.ndim == 2 and d.shape[1] == 3 else d.shape)
