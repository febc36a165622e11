#!/usr/bin/env python3
"""Dev tool (GPU box): differential fuzzing of the HIP path against the CPU oracle over random
configurations (map size, columns, headings, field of view, radius, step, camera height, image
height, world-unit type, the three unpinned switches, both BoundsError policies, auto-reset).

    python tools/fuzz_parity.py [configs] [seed] [top|split|flat]   # "top": every configuration renders the top view;
                                                                   # "split": ... with a geometry of the unit store kernels;
                                                                   # "flat": ... of the flat store kernel (any pu >= 9), and any
                                                                   #         camera height from 24 rows (the flat fill kernel)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import raycastworlds_jl_amd as RCW
from helpers import assert_state_equal
from oracle import oracle as O

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
always_top = len(sys.argv) > 3 and sys.argv[3] in ("top", "split", "flat")
split_geometry = len(sys.argv) > 3 and sys.argv[3] == "split"   # geometries of the unit store kernels (units of 256 / 128 / 64 / 32 rows)
flat_geometry = len(sys.argv) > 3 and sys.argv[3] == "flat"     # geometries of the flat store kernel; the two-kernel form is asked for
step_geometry = len(sys.argv) > 3 and sys.argv[3] == "step"     # what the one-launch step takes (a camera view of 256 k / 128 / 64 rows, no top view), asked for in every case:
                                                                # 1 .. 1,500 view columns (a wavefront per agent, a workgroup per agent, the table's tail), maps of up to 40 x 40 tiles
one_launch_steps = 0
two_kernel_forms = 0                                            # (rcw_set_top_view_form: by default it is taken only from 256 MiB a step)
fails = 0
O.set_num_threads(8)
for c in range(n_cfg):
    if c % 10 == 0:
        print(f"config {c} ...", flush=True)
    T64 = bool(rng.integers(0, 2))
    radius = float(rng.choice([1 / 8, 0.05, 0.2, 0.3, 0.49]))
    inc = float(rng.choice([1 / 8, 1 / 16, 0.1, 0.03, radius]))
    kw = dict(height_tile_map_tu=int(rng.integers(4, 24)), width_tile_map_tu=int(rng.integers(4, 24)),
              num_rays=int(rng.choice([1, 2, 7, 64, 100, 256, 333])), num_directions=int(rng.choice([4, 16, 36, 128, 360])),
              player_radius_wu=radius, position_increment_wu=min(inc, radius),
              semi_field_of_view_wu=float(rng.choice([2 / 3, 0.25, 1.0, 1.7])),
              camera_height_tile_wu=float(rng.choice([1.0, 0.5, 2.5])),
              height_camera_view_pu=int(rng.choice([256, 64, 100, 37, 512, 128, 84, 768])),
              dda_tie_break=int(rng.integers(0, 2)), dda_distance=int(rng.integers(0, 2)),
              normalize_mode=int(rng.integers(0, 2)), out_of_bounds=int(rng.integers(0, 2)),
              auto_reset=bool(rng.integers(0, 2)), render_top_view=bool(rng.integers(0, 4) == 0) or always_top,
              pu_per_tu=int(rng.choice([4, 8, 13, 32, 40, 52] if always_top else [4, 8, 13, 32])))
    R = str(rng.choice(["Float32", "Float64", "Int32", "Int64"]))
    B = int(rng.integers(1, 40))
    if step_geometry:
        kw.update(height_camera_view_pu=int(rng.choice([256, 256, 256, 64, 128, 512, 768, 1024, 2048])), render_top_view=False, num_rays=int(rng.choice([1, 7, 64, 100, 255, 256, 257, 333, 512, 700, 1024, 1100, 1500])),
                  height_tile_map_tu=int(rng.integers(3, 41)), width_tile_map_tu=int(rng.integers(3, 41)))
        B = int(rng.choice([1, 3, 4, 5, 17, 64, 130]))
    if split_geometry:
        # image heights of 32 m rows with tiles that divide the store kernel's unit (256, 128, 64 or 32 rows)
        pu = int(rng.choice([8, 16, 32, 32, 64, 128]))
        kw["pu_per_tu"] = pu
        m = int(rng.integers(2, 25))
        H = max(4, (32 * m) // pu) if pu <= 64 else int(rng.integers(4, 7)) // 2 * 2
        kw["height_tile_map_tu"] = H
        kw["width_tile_map_tu"] = int(rng.integers(4, 10 if pu >= 64 else 20))
        B = int(rng.integers(1, 12 if pu >= 64 else 40))
    if flat_geometry:
        pu = int(rng.integers(9, 61))
        kw["pu_per_tu"] = pu
        H = int(rng.integers(4, 20))
        if (H * pu) % 4:                                   # image height a multiple of 4
            H += (4 - H % 4) % 4 if pu % 2 else (2 - H % 2) % 2
        kw["height_tile_map_tu"] = max(H, 4)
        kw["width_tile_map_tu"] = int(rng.integers(4, 14))
        kw["height_camera_view_pu"] = int(rng.choice([int(rng.integers(24, 700)), int(rng.integers(24, 48)), 84, 100, 250, 300, 333, 40]))
        B = int(rng.integers(1, 30))
    seed = int(rng.integers(0, 2**31))
    okw = {k: v for k, v in kw.items()}
    okw["auto_reset"] = int(kw["auto_reset"]); okw["render_top_view"] = int(kw["render_top_view"])
    okw["reward_type"] = ["Float32", "Float64", "Int32", "Int64"].index(R)
    if T64:
        okw["world_unit_bits"] = 64
        for k in ("player_radius_wu", "position_increment_wu", "semi_field_of_view_wu", "camera_height_tile_wu"):
            okw[k + "_f64"] = float(kw[k])
    try:
        env = RCW.SingleRoomModule.SingleRoom(batch=B, seed=seed, T="Float64" if T64 else "Float32", R=R, **kw)
        orc = O.OracleBatch(B, seed=seed, **okw)
        assert_state_equal(env, orc, rays=True, where="create")
        if kw["render_top_view"] and (split_geometry or flat_geometry or rng.integers(0, 2)):
            try:
                env.set_top_view_form("two-kernels", runs=int(rng.integers(0, 4)))
                two_kernel_forms += 1
            except Exception:                                  # the geometry does not take it: the automatic form stays
                pass
        if step_geometry or rng.integers(0, 2):               # the one-launch step (these batches are below where the rule takes it by itself)
            try:
                env.set_step_form("one-launch")
                one_launch_steps += 1
            except Exception:                                  # another camera height / a top view: the two launches stay
                pass
        if rng.integers(0, 2):
            # arbitrary injected poses: uniform, exactly on tile boundaries, a hair off them, tile centres,
            # possibly inside the goal tile (a ray that starts inside an obstacle)
            H, W = kw["height_tile_map_tu"], kw["width_tile_map_tu"]
            real = np.float64 if T64 else np.float32
            def coords(n, hi):
                kind = rng.integers(0, 4, n)
                u = rng.uniform(1.0, hi - 1.0, n)
                k = rng.integers(1, hi - 1, n).astype(np.float64)
                eps = rng.choice([1e-7, -1e-7, 1e-12, 3e-5], n)
                out = np.where(kind == 0, u, np.where(kind == 1, k, np.where(kind == 2, k + eps, k + 0.5)))
                out = np.clip(out, 1.0, np.nextafter(real(hi - 1), real(0))).astype(real)
                return np.where(out < 1, real(1), out)
            pos = np.stack([coords(B, H), coords(B, W)], axis=1).astype(real)
            goal = np.stack([rng.integers(2, H, B), rng.integers(2, W, B)], axis=1).astype(np.int32)
            d = rng.integers(0, kw["num_directions"], B).astype(np.int32)
            env.set_state(goal, pos, d)
            orc.set_state(goal, pos, d)
            assert_state_equal(env, orc, rays=True, where="set_state with arbitrary poses")
        for s in range(int(rng.integers(5, 60))):
            if rng.integers(0, 12) == 0:                       # a masked reset with a fresh seed now and then
                mask = (rng.random(B) < 0.4).astype(np.uint8)
                sd = int(rng.integers(0, 2**31))
                RCW.reset_(env, mask=mask, seed=sd)
                orc.reset(mask=mask, seed=sd)
            a = rng.integers(1, 5, B).astype(np.uint8)
            RCW.act_(env, a)
            assert orc.step(a) == 0
            try:
                env.sync()
            except IndexError:
                np.testing.assert_array_equal(env.world.status, orc.status)
                env.clear_error(); orc.clear_status()
        assert_state_equal(env, orc, rays=True, where="rollout")
        np.testing.assert_array_equal(env.world.episode, orc.episode)
        if kw["render_top_view"]:
            np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        env.close(); orc.close()
    except Exception as e:   # noqa: BLE001
        fails += 1
        print(f"config {c} FAILED: T64={T64} R={R} B={B} seed={seed} {kw}\n   {type(e).__name__}: {str(e)[:300]}")
        if fails >= 5:
            break
print(f"{n_cfg} random configurations ({two_kernel_forms} with the two-kernel top view asked for and taken, {one_launch_steps} with the one-launch step), {fails} mismatches")
sys.exit(1 if fails else 0)
