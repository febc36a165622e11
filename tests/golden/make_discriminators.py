#!/usr/bin/env python3
"""Builds the DISCRIMINATING inputs of the parity-pin kit: tests/golden/discriminators.json and
julia/discriminator_cases.tsv.

Three pieces of the path's arithmetic live in un-vendored Julia packages (RayCaster.cast_ray,
StaticArrays' normalize; include/rcw.h "UNPINNED") and each has two plausible readings, selectable by one
rcw_config field:  dda_tie_break (`<` / `<=`),  dda_distance (side - delta / pre-increment),
normalize_mode (inv(norm) * v  /  v / norm).  Almost every pose renders identically under both readings,
so a random comparison with the real package would say little.  This script finds poses where the two
readings DIFFER in what the reference exposes (ray_hit_dimension, ray_stop_position_tu, ray_distance_wu,
ray_directions_wu, camera_view), and records both candidate outputs, computed by the CPU oracle:

  * dda_tie_break: constructed — a diagonal heading from a tile centre with an odd fan puts the centre ray
    through exact tile corners, so side_x == side_y at every step (exact in Float32);
  * dda_distance, normalize_mode: found by search over random poses (last-ulp effects that flip
    floor(height_line) or a stop tile).

One run of julia/make_reference_fixtures.jl on these cases (with the real RayCastWorlds.jl) then decides each
switch; tests/test_reference_fixtures.py reads its output.  Until then: parity unpinned.

    python tests/golden/make_discriminators.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

SWITCHES = ("dda_tie_break", "dda_distance", "normalize_mode")


def outputs(cfg, goal, pos, d, **switch):
    """What the reference exposes after cast_rays! + update_camera_view! for ONE injected state."""
    orc = O.OracleBatch(1, out_of_bounds=1, **cfg, **switch)
    orc.set_state([goal], [pos], [d])
    out = dict(
        ray_hit_dimension=orc.ray_dim[0].tolist(),
        ray_stop_position_tu=orc.ray_stop[0].reshape(-1).tolist(),            # (i, j) per ray
        ray_distance_bits=orc.ray_dist[0].view(np.uint32).tolist(),
        ray_direction_bits=orc.ray_dirs[0].reshape(-1).view(np.uint32).tolist(),
        height_line_pu=orc.col_height[0].tolist(),                            # by image column k = N - i + 1
        colour_id=orc.col_colour[0].tolist(),
    )
    orc.close()
    return out


def differing_fields(a, b):
    return [k for k in a if a[k] != b[k]]


def search(cfg, switch, want_frame_difference, seed, batch=8192, rounds=40):
    """Random poses whose outputs differ between the two readings of `switch`."""
    rng = np.random.default_rng(seed)
    H, W = cfg["height_tile_map_tu"], cfg["width_tile_map_tu"]
    found = []
    for _ in range(rounds):
        goal = np.stack([rng.integers(2, H, batch), rng.integers(2, W, batch)], axis=1).astype(np.int32)
        # positions on the 1/64 lattice (every state reachable in 1/8 steps from a tile centre along the 128
        # headings is a Float32 near such points; the lattice keeps the cases short to write down), at least
        # one radius away from the wall ring
        pos = (rng.integers(72, 64 * (min(H, W) - 1) - 8, (batch, 2)) / 64.0).astype(np.float32)
        d = rng.integers(0, 128, batch).astype(np.int32)
        # keep the player out of the goal tile's neighbourhood (an un-reachable pose would prove nothing)
        tile = np.floor(pos).astype(np.int32) + 1
        ok = (np.abs(tile - goal).max(axis=1) > 1)
        a = O.OracleBatch(batch, render=False, out_of_bounds=1, **cfg, **{switch: 0})
        b = O.OracleBatch(batch, render=False, out_of_bounds=1, **cfg, **{switch: 1})
        a.set_state(goal, pos, d); b.set_state(goal, pos, d)
        frame_diff = ((a.col_height != b.col_height) | (a.col_colour != b.col_colour)).any(axis=1)
        ray_diff = ((a.ray_dist.view(np.uint32) != b.ray_dist.view(np.uint32)).any(axis=1) |
                    (a.ray_dirs.view(np.uint32) != b.ray_dirs.view(np.uint32)).any(axis=(1, 2)) |
                    (a.ray_dim != b.ray_dim).any(axis=1))
        pick = np.nonzero(ok & (frame_diff if want_frame_difference else (ray_diff & ~frame_diff)))[0]
        a.close(); b.close()
        for k in pick[:4]:
            found.append((goal[k].tolist(), [float(pos[k, 0]), float(pos[k, 1])], int(d[k])))
        if len(found) >= 3:
            break
    return found[:3]


def main():
    cfg1 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    cfg2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
    cases = []

    def add(name, switch, cfg, goal, pos, d, why):
        cand = [outputs(cfg, goal, pos, d, **{switch: v}) for v in (0, 1)]
        diff = differing_fields(cand[0], cand[1])
        assert diff, f"{name}: the two readings of {switch} agree — not a discriminator"
        cases.append(dict(name=name, switch=switch, config=cfg, goal=goal, position=pos, direction=d, why=why,
                          differs_in=diff, candidates={"0": cand[0], "1": cand[1]}))
        print(f"{name:34s} {switch:15s} differs in {diff}")

    # ---- dda_tie_break: constructed ----
    odd = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=65)
    for heading, goal in ((16, [2, 2]), (48, [7, 2]), (80, [7, 7]), (112, [2, 7])):
        add(f"tie_diagonal_heading_{heading}", "dda_tie_break", odd, goal, [4.5, 4.5], heading,
            "heading on a diagonal (cos == sin in Float32), player on a tile centre, 65 rays: the centre ray "
            "(LinRange element 33 = 0.5*first + 0.5*last = the heading) has |dx| == |dy| and equal fractional "
            "offsets, so side_x == side_y at EVERY step; `<` steps y first and stops in a y-face (hit_dimension 2), "
            "`<=` steps x first (hit_dimension 1): the centre column changes shade")
    # ---- dda_distance / normalize_mode: searched ----
    for switch, seed in (("dda_distance", 101), ("normalize_mode", 202)):
        for cfg, tag in ((cfg1, "n64"), (cfg2, "n256")):
            for want_frame, kind in ((True, "frame"), (False, "rays_only")):
                hits = search(cfg, switch, want_frame, seed)
                seed += 1
                for n, (goal, pos, d) in enumerate(hits[:2 if want_frame else 1]):
                    add(f"{switch}_{tag}_{kind}_{n}", switch, cfg, goal, pos, d,
                        ("a last-ulp difference of the two readings flips floor(height_line) or a stop tile: camera_view differs"
                         if want_frame else
                         "the two readings give different Float32 bits in ray_distance_wu / ray_directions_wu; camera_view agrees"))

    out = dict(
        _about="Discriminating inputs for the UNPINNED switches of include/rcw.h, with both candidate outputs per case "
               "computed by the CPU oracle (tests/golden/make_discriminators.py).  Floats are IEEE-754 Float32 bit "
               "patterns.  height_line_pu / colour_id are indexed by image column k = N - i + 1 (SR:431) and determine "
               "camera_view completely (SR:431-440).  Colour ids: 0 wall_dim_1, 1 wall_dim_2, 2 goal_dim_1, 3 goal_dim_2.",
        switches={"dda_tie_break": {"0": "x first when side_x < side_y", "1": "x first when side_x <= side_y"},
                  "dda_distance": {"0": "side - delta after the step", "1": "side before the increment"},
                  "normalize_mode": {"0": "inv(norm(v)) * v", "1": "v / norm(v)"}},
        cases=cases,
        # not switch discriminators: inputs whose top_view pins SimpleDraw's rasterisers on a pixel scale that is not a power of two
        rasteriser_cases=[dict(
            name="top_view_pu12_rasterisers",
            config=dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64, pu_per_tu=12),
            goal=[6, 3], position=[3.71875, 5.40625], direction=23,
            why="not a switch discriminator: pins SimpleDraw's Line / Circle (top_view) on a pixel scale that is not a power of two — "
                "12 pixels a tile: wu_to_pu end points, Bresenham lines and the midpoint circle (radius_pu = 2) on a 96 x 96 image, "
                "where the engine's flat store kernel (any pixel scale from 9) renders; compared through <name>.top_view.u32"),
            dict(
            name="top_view_pu13_rasterisers",
            config=dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64, pu_per_tu=13),
            goal=[3, 6], position=[5.28125, 2.65625], direction=101,
            why="not a switch discriminator: SimpleDraw's Line / Circle at 13 pixels a tile — a scale that is not a multiple of 4, "
                "so a lane's four pixels straddle tiles and the 104 x 104 image is 42.25 chunks (the flat store kernel's STRADDLE "
                "path); wu_to_pu(x) = ceil(13 x) lands off the Float32 grid's round numbers; compared through <name>.top_view.u32")])
    with open(os.path.join(HERE, "discriminators.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    # the inputs alone, in a form julia/make_reference_fixtures.jl parses with Base only
    with open(os.path.join(ROOT, "julia", "discriminator_cases.tsv"), "w") as f:
        f.write("# name\tH\tW\tnum_rays\tgoal_i\tgoal_j\tx_bits\ty_bits\tdirection_au\t[pu_per_tu = 32]   (Float32 bit patterns; "
                "written by tests/golden/make_discriminators.py)\n")
        for c in cases + out["rasteriser_cases"]:
            x, y = np.array(c["position"], dtype=np.float32).view(np.uint32).tolist()
            cfg = c["config"]
            f.write(f"{c['name']}\t{cfg['height_tile_map_tu']}\t{cfg['width_tile_map_tu']}\t{cfg['num_rays']}\t"
                    f"{c['goal'][0]}\t{c['goal'][1]}\t{x}\t{y}\t{c['direction']}"
                    + (f"\t{cfg['pu_per_tu']}" if "pu_per_tu" in cfg else "") + "\n")
    print(len(cases), "cases written")


if __name__ == "__main__":
    main()
