"""Two independent restatements of the reference (C: oracle/rcw_oracle.c, Python:
oracle/pyref.py) must agree bit for bit.  CPU only, small cases."""
import numpy as np
import pytest

from oracle import pyref


def _compare(orc, w, where):
    np.testing.assert_array_equal(orc.position[0], np.array(w.pos, dtype=np.float32), err_msg=where)
    assert orc.direction[0] == w.dir, where
    assert orc.reward[0] == w.reward and bool(orc.done[0]) == w.done, where
    np.testing.assert_array_equal(orc.col_height[0], np.array(w.col_height, dtype=np.int32), err_msg=where)
    np.testing.assert_array_equal(orc.col_colour[0], np.array(w.col_colour, dtype=np.uint8), err_msg=where)
    np.testing.assert_array_equal(orc.camera_view[0], w.camera_view, err_msg=where)
    hits = np.array([(h[0], h[1]) for h in w.ray_hits], dtype=np.int64)
    np.testing.assert_array_equal(orc.ray_stop[0], hits, err_msg=where)
    np.testing.assert_array_equal(orc.ray_dim[0], np.array([h[2] for h in w.ray_hits], dtype=np.int64), err_msg=where)
    dist = np.array([h[3] for h in w.ray_hits], dtype=np.float32)
    np.testing.assert_array_equal(orc.ray_dist[0].view(np.uint32), dist.view(np.uint32), err_msg=where)
    dirs = np.array(w.ray_dirs, dtype=np.float32)
    np.testing.assert_array_equal(orc.ray_dirs[0].view(np.uint32), dirs.view(np.uint32), err_msg=where)


@pytest.mark.parametrize("H,W,N,steps,variant", [
    (8, 8, 64, 60, (0, 0, 0)),
    (8, 16, 48, 40, (0, 0, 0)),
    (16, 16, 33, 30, (0, 0, 0)),
    (8, 8, 32, 30, (1, 0, 0)),
    (8, 8, 32, 30, (0, 1, 0)),
    (8, 8, 32, 30, (0, 0, 1)),
    (8, 8, 1, 10, (0, 0, 0)),       # LinRange of length 1: lendiv = max(len-1, 1)
])
def test_rollout_bit_exact(oracle, H, W, N, steps, variant):
    tie, dist, norm = variant
    orc = oracle.OracleBatch(1, seed=H * 1000 + N, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N,
                             dda_tie_break=tie, dda_distance=dist, normalize_mode=norm, out_of_bounds=0)
    w = pyref.World(H=H, W=W, num_rays=N, tie_le=bool(tie), dist_pre=bool(dist), normalize_divide=bool(norm))
    np.testing.assert_array_equal(orc.directions, np.array(w.directions, dtype=np.float32))
    w.set_state(orc.goal[0], orc.position[0], orc.direction[0])
    _compare(orc, w, "after set_state")
    rng = np.random.default_rng(N)
    for s in range(steps):
        a = int(rng.integers(1, 5))
        try:
            w.step(a)
        except IndexError:
            # the reference's BoundsError: the oracle flags the agent and leaves it untouched
            orc.step([a])
            assert orc.status[0] == -5
            orc.clear_status()
            w.cast_rays(); w.update_camera_view()
            continue
        assert orc.step([a]) == 0
        _compare(orc, w, f"step {s} action {a}")


def test_all_headings_ray_fan(oracle):
    cfg = oracle.default_config(num_rays=17)
    w = pyref.World(num_rays=17)
    for d in range(128):
        w.dir = d
        fan = oracle.ray_fan(cfg, oracle.direction_table(128)[d])
        np.testing.assert_array_equal(fan.view(np.uint32), np.array(w.ray_fan(), dtype=np.float32).view(np.uint32))


def test_top_view_two_restatements_agree(oracle):
    """update_top_view! SR:446-483 restated twice (C and Python) — same pixels."""
    for H, W, N, pu in ((8, 8, 64, 32), (8, 16, 40, 16), (6, 7, 33, 9)):
        orc = oracle.OracleBatch(1, seed=H + N, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N,
                                 render_top_view=1, pu_per_tu=pu, out_of_bounds=1)
        w = pyref.World(H=H, W=W, num_rays=N)
        w.set_state(orc.goal[0], orc.position[0], orc.direction[0])
        rng = np.random.default_rng(pu)
        for s in range(25):
            a = int(rng.integers(1, 5))
            orc.step([a])
            try:
                w.step(a)
            except IndexError:          # reference BoundsError: with out_of_bounds=1 the oracle blocks the move
                w.cast_rays(); w.update_camera_view()
        w.update_top_view(pu)
        np.testing.assert_array_equal(w.top_view, orc.top_view[0])


@pytest.mark.parametrize("H,W,N,steps", [(8, 8, 64, 50), (8, 16, 40, 40), (7, 9, 21, 30)])
def test_float64_rollout_bit_exact(oracle, H, W, N, steps):
    """T = Float64: the Python restatement with numpy Float64 scalars vs the C oracle compiled with
    -DORC_REAL64, bit for bit (positions, distances, ray directions as doubles)."""
    orc = oracle.OracleBatch(1, seed=7 * H + N, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N,
                             world_unit_bits=64, out_of_bounds=1, render_top_view=1, pu_per_tu=16)
    try:
        w = pyref.World(H=H, W=W, num_rays=N, T=np.float64)
        assert orc.position.dtype == np.float64
        np.testing.assert_array_equal(orc.directions, np.array(w.directions, dtype=np.float64))
        w.set_state(orc.goal[0], orc.position[0], orc.direction[0])
        rng = np.random.default_rng(N)
        for s in range(steps):
            a = int(rng.integers(1, 5))
            orc.step([a])
            try:
                w.step(a)
            except IndexError:
                w.cast_rays(); w.update_camera_view()
            np.testing.assert_array_equal(orc.position[0], np.array(w.pos, dtype=np.float64), err_msg=f"step {s}")
            assert orc.direction[0] == w.dir
            np.testing.assert_array_equal(orc.col_height[0], np.array(w.col_height, dtype=np.int32))
            np.testing.assert_array_equal(orc.col_colour[0], np.array(w.col_colour, dtype=np.uint8))
            dist = np.array([h[3] for h in w.ray_hits], dtype=np.float64)
            np.testing.assert_array_equal(orc.ray_dist[0].view(np.uint64), dist.view(np.uint64))
            np.testing.assert_array_equal(orc.ray_dirs[0].view(np.uint64), np.array(w.ray_dirs, dtype=np.float64).view(np.uint64))
        np.testing.assert_array_equal(orc.camera_view[0], w.camera_view)
        w.update_top_view(16)
        np.testing.assert_array_equal(w.top_view, orc.top_view[0])
    finally:
        orc.close()


def test_float64_differs_from_float32_where_it_should(oracle):
    """convert(Float64, 2/3) is not Float32(2/3) widened: the two worlds cast different fans."""
    o32 = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    o64 = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64, world_unit_bits=64)
    for o in (o32, o64):
        o.set_state([[2, 2]], [[4.5, 4.5]], [5])
    assert o64.ray_dirs.dtype == np.float64 and o32.ray_dirs.dtype == np.float32
    assert not np.array_equal(o32.ray_dirs[0].astype(np.float64), o64.ray_dirs[0])
    np.testing.assert_allclose(o32.ray_dirs[0], o64.ray_dirs[0], atol=1e-6)
    np.testing.assert_array_equal(o32.col_height, o64.col_height)      # same picture at this pose
    assert o64.directions[32, 0] == 6.123233995736766e-17 and o64.directions[32, 1] == 1.0


@pytest.mark.parametrize("bits", [32, 64])
def test_arbitrary_poses_two_restatements_agree(oracle, bits):
    """Injected poses on tile boundaries, a hair off them, inside the goal tile, for every 7th heading:
    the edge cases of cast_ray (zero side distance, ties, a ray that starts inside an obstacle)."""
    T = np.float64 if bits == 64 else np.float32
    rng = np.random.default_rng(bits)
    H, W, N = 9, 7, 16
    orc = oracle.OracleBatch(1, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N, world_unit_bits=bits)
    try:
        w = pyref.World(H=H, W=W, num_rays=N, T=T)
        for trial in range(60):
            kind = rng.integers(0, 4, 2)
            vals = []
            for axis, hi in enumerate((H, W)):
                k = float(rng.integers(1, hi - 1))
                v = [rng.uniform(1.0, hi - 1.0), k, k + float(rng.choice([1e-7, -1e-7, 3e-5])), k + 0.5][kind[axis]]
                vals.append(min(max(T(v), T(1)), np.nextafter(T(hi - 1), T(0))))
            goal = (int(rng.integers(2, H)), int(rng.integers(2, W)))
            d = int(rng.integers(0, 128))
            pos = np.array([vals], dtype=T)
            orc.set_state([goal], pos, [d])
            w.set_state(goal, pos[0], d)
            np.testing.assert_array_equal(orc.col_height[0], np.array(w.col_height, dtype=np.int32), err_msg=f"trial {trial}")
            np.testing.assert_array_equal(orc.col_colour[0], np.array(w.col_colour, dtype=np.uint8))
            np.testing.assert_array_equal(orc.ray_stop[0], np.array([(h[0], h[1]) for h in w.ray_hits], dtype=np.int64))
            np.testing.assert_array_equal(orc.ray_dim[0], np.array([h[2] for h in w.ray_hits], dtype=np.int64))
            np.testing.assert_array_equal(orc.ray_dist[0], np.array([h[3] for h in w.ray_hits], dtype=T))
            np.testing.assert_array_equal(orc.camera_view[0], w.camera_view)
    finally:
        orc.close()


@pytest.mark.parametrize("rt,R", [(1, np.float64), (2, np.int32), (3, np.int64)], ids=["Float64", "Int32", "Int64"])
def test_reward_type_R(oracle, rt, R):
    """R of SingleRoom(; R = ...) SR:266 in both restatements: reward = zero(R) / goal_reward = one(R) (SR:81-82),
    element type included.  A walk into the goal, a turn, and back (the hand-derived dynamics case)."""
    orc = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=16, reward_type=rt)
    w = pyref.World(H=8, W=8, num_rays=16, R=R)
    orc.set_state([[6, 5]], [[4.5, 4.5]], [0])
    w.set_state((6, 5), (4.5, 4.5), 0)
    seen = set()
    for a in (1, 1, 1, 1, 3, 4, 1, 2, 1):
        orc.step([a]); w.act(a)
        assert orc.reward.dtype == R and type(w.reward) is R
        assert orc.reward[0] == w.reward and bool(orc.done[0]) == w.done
        seen.add((int(w.reward), w.done))
    assert seen == {(0, False), (1, True)}
