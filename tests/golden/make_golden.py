#!/usr/bin/env python3
"""Generates the golden rollouts in this directory from the CPU oracle (oracle/rcw_oracle.c).

The reference (Julia) cannot run in this pipeline and holds no fixtures of its own for this
path, so these vectors pin the ORACLE's current behaviour (a change in the oracle or in the
HIP path shows up as a diff against committed data); what ties the oracle to the reference
text is hand_derived.json (derived on paper from src/single_room.jl) and oracle/pyref.py.

    python tests/golden/make_golden.py        # rewrites *.npz next to this file
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402

CASES = {
    # name: (config overrides, agents, steps, seed)
    "cfg1_8x8_n64": (dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64), 4, 200, 11),
    "cfg2_8x8_n256": (dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256), 4, 80, 15),
    "default_8x16_n512": (dict(), 2, 60, 12),
    "cfg3_16x16_n512": (dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=512), 2, 40, 13),
    "cfg4_16x16_n256": (dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=256), 4, 60, 16),
    "cfg5_32x32_n1024": (dict(height_tile_map_tu=32, width_tile_map_tu=32, num_rays=1024), 2, 24, 14),
}


def frame_crcs(frames):
    return np.array([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in frames], dtype=np.uint32)


def make(name):
    cfg, B, steps, seed = CASES[name]
    # treat-empty policy so that the rollout never stops on the reference's BoundsError quirk
    orc = O.OracleBatch(B, seed=seed, out_of_bounds=1, **cfg)
    rng = np.random.default_rng(seed)
    actions = rng.integers(1, 5, (steps, B)).astype(np.uint8)
    out = dict(
        seed=np.int64(seed), actions=actions,
        goal0=orc.goal.copy(), pos0=orc.position.copy(), dir0=orc.direction.copy(),
        tile_map0=orc.tile_map_chunks(),
        col_h0=orc.col_height.copy(), col_c0=orc.col_colour.copy(), crc0=frame_crcs(orc.camera_view),
        pos=np.zeros((steps, B, 2), np.float32), dir=np.zeros((steps, B), np.int32),
        reward=np.zeros((steps, B), np.float32), done=np.zeros((steps, B), np.uint8),
        col_h=np.zeros((steps, B, orc.N), np.int32), col_c=np.zeros((steps, B, orc.N), np.uint8),
        crc=np.zeros((steps, B), np.uint32),
    )
    for s in range(steps):
        assert orc.step(actions[s]) == 0
        out["pos"][s] = orc.position; out["dir"][s] = orc.direction
        out["reward"][s] = orc.reward; out["done"][s] = orc.done
        out["col_h"][s] = orc.col_height; out["col_c"][s] = orc.col_colour
        out["crc"][s] = frame_crcs(orc.camera_view)
    out["config_keys"] = np.array(sorted(cfg.keys()))
    out["config_vals"] = np.array([cfg[k] for k in sorted(cfg.keys())], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written")


if __name__ == "__main__":
    # `make_golden.py name...` rewrites only those cases; without arguments, all of them
    for n in (sys.argv[1:] or CASES):
        make(n)
