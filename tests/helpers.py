"""Shared comparison helpers for the parity tests (HIP path vs CPU oracle)."""
import numpy as np

# BASELINE.json configs (map H x W, view columns); batch sizes are chosen per test.
CFG1 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
CFG2 = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=256)
CFG3 = dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=512)
CFG4 = dict(height_tile_map_tu=16, width_tile_map_tu=16, num_rays=256)
CFG5 = dict(height_tile_map_tu=32, width_tile_map_tu=32, num_rays=1024)
REFERENCE_DEFAULT = dict()   # 8 x 16, 512 rays: what test/runtests.jl:19 exercises

# tolerance north_star states for player_position_wu; everything else is bit-exact
POSITION_ATOL = 1e-6


def assert_state_equal(env, orc, frames=True, rays=False, where=""):
    """Compare every observable of the HIP engine with the oracle's."""
    w = env.world
    np.testing.assert_array_equal(w.player_direction_au, orc.direction, err_msg=f"direction {where}")
    np.testing.assert_array_equal(w.goal_position, orc.goal, err_msg=f"goal {where}")
    pos = w.player_position_wu
    np.testing.assert_allclose(pos, orc.position, rtol=0, atol=POSITION_ATOL, err_msg=f"position {where}")
    # the engine is in fact bit-exact on positions too
    assert pos.dtype == orc.position.dtype, (pos.dtype, orc.position.dtype)
    bits = np.uint64 if pos.dtype == np.float64 else np.uint32
    np.testing.assert_array_equal(pos.view(bits), orc.position.view(bits), err_msg=f"position bits {where}")
    np.testing.assert_array_equal(w.reward, orc.reward, err_msg=f"reward {where}")
    np.testing.assert_array_equal(w.done.astype(np.uint8), orc.done, err_msg=f"done {where}")
    np.testing.assert_array_equal(w.tile_map_chunks, orc.tile_map_chunks(), err_msg=f"tile_map {where}")
    h, c = env.columns()
    np.testing.assert_array_equal(h, orc.col_height, err_msg=f"height_line_pu {where}")
    np.testing.assert_array_equal(c, orc.col_colour, err_msg=f"colour id {where}")
    if frames:
        np.testing.assert_array_equal(env.camera_view_host(), orc.camera_view, err_msg=f"camera_view {where}")
    if rays:
        stop, dim, dist, dirs = w.rays()
        np.testing.assert_array_equal(stop, orc.ray_stop, err_msg=f"ray_stop {where}")
        np.testing.assert_array_equal(dim, orc.ray_dim, err_msg=f"ray_dim {where}")
        assert dist.dtype == orc.ray_dist.dtype
        np.testing.assert_array_equal(dist.view(bits), orc.ray_dist.view(bits), err_msg=f"ray_dist {where}")
        np.testing.assert_array_equal(dirs.view(bits), orc.ray_dirs.view(bits), err_msg=f"ray_dirs {where}")
