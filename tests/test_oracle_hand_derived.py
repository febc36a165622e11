"""Pins the CPU oracle (and the independent pyref restatement) to known answers derived by
hand from the reference text: tests/golden/hand_derived.json.  CPU only."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "hand_derived.json")) as f:
    KAT = json.load(f)
COL = KAT["colours"]
CID = {"wall_dim_1": 0, "wall_dim_2": 1, "goal_dim_1": 2, "goal_dim_2": 3}


def _ring(H=8, W=8, goal=None):
    m = np.zeros((H, W), dtype=bool)
    m[0, :] = m[-1, :] = True
    m[:, 0] = m[:, -1] = True
    if goal is not None:
        m[goal[0] - 1, goal[1] - 1] = True
    return m


@pytest.mark.parametrize("case", KAT["render"], ids=lambda c: c["name"][:40])
def test_render_known_answers(oracle, case):
    N = case["num_rays"]
    orc = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=N)
    orc.set_state([case["goal"]], [case["position"]], [case["direction"]])
    h, c, frame = orc.col_height[0], orc.col_colour[0], orc.camera_view[0]
    if "all_columns" in case:
        a = case["all_columns"]
        assert (h == a["h"]).all()
        assert (c == CID[a["colour"]]).all()
        assert (orc.ray_dim[0] == a["hit_dimension"]).all()
        if "stop_i" in a:
            assert (orc.ray_stop[0][:, 0] == a["stop_i"]).all()
        if "stop_j" in a:
            assert (orc.ray_stop[0][:, 1] == a["stop_j"]).all()
        if "ray1_stop_j" in case:
            assert orc.ray_stop[0][0, 1] == case["ray1_stop_j"]
        if "rows_1based" in case:
            r = case["rows_1based"]
            for k in range(N):
                col = frame[k]
                assert (col[r["ceiling"][0] - 1:r["ceiling"][1]] == COL["ceiling"]).all()
                assert (col[r["colour"][0] - 1:r["colour"][1]] == COL[a["colour"]]).all()
                assert (col[r["floor"][0] - 1:r["floor"][1]] == COL["floor"]).all()
    for spec in case.get("columns_1based", []):
        k0, k1 = spec["k"]
        assert (h[k0 - 1:k1] == spec["h"]).all(), (spec, h[k0 - 1:k1])
        assert (c[k0 - 1:k1] == CID[spec["colour"]]).all(), (spec, c[k0 - 1:k1])
    if case["name"].startswith("goal tile"):
        # the red band is exactly columns 24..41
        red = np.nonzero(c >= 2)[0] + 1
        assert red.min() == 24 and red.max() == 41 and len(red) == 18


@pytest.mark.parametrize("case", KAT["dynamics"], ids=lambda c: c["name"][:40])
def test_dynamics_known_answers(oracle, case):
    orc = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    orc.set_state([case["goal"]], [case["position"]], [case["direction"]])
    for (action, count), expect in zip(case["actions"], case["after_each_block"]):
        for _ in range(count):
            assert orc.step([action]) == 0
        if "position" in expect:
            np.testing.assert_array_equal(orc.position[0], np.array(expect["position"], dtype=np.float32))
        if "reward" in expect:
            assert orc.reward[0] == np.float32(expect["reward"])
        if "done" in expect:
            assert bool(orc.done[0]) == expect["done"]
        if "direction" in expect:
            assert orc.direction[0] == expect["direction"]


@pytest.mark.parametrize("case", KAT["cast_ray"], ids=lambda c: c["name"][:30])
def test_cast_ray_known_answers(oracle, case):
    e = case["expect"]
    i, j, dim, dist = oracle.cast_ray(_ring(), *case["start"], *[np.float32(v) for v in case["dir"]])
    assert (i, j, dim) == (e["i"], e["j"], e["dim"])
    assert abs(float(dist) - e["dist"]) <= e.get("dist_tol", 0.0)


@pytest.mark.parametrize("case", KAT["collision"], ids=lambda c: c["name"][:30])
def test_collision_known_answers(oracle, case):
    if case["layer"] == "wall":
        layer = _ring()
    else:
        layer = np.zeros((8, 8), dtype=bool)
        layer[case["goal"][0] - 1, case["goal"][1] - 1] = True
    got = oracle.is_player_colliding(layer, np.float32(case["position"][0]), np.float32(case["position"][1]),
                                     np.float32(0.125))
    assert got == case["expect"]


def test_direction_table_facts(oracle):
    """dir[0] = (1, 0) with an exact zero, the only one for nd = 128."""
    d = oracle.direction_table(128)
    assert d[0, 0] == 1.0 and d[0, 1] == 0.0
    assert np.count_nonzero(d == 0.0) == 1
    assert d[64, 0] == -1.0 and d[64, 1] == np.float32(1.2246469e-16)
    assert d[32, 0] == np.float32(6.123234e-17) and d[32, 1] == 1.0
    np.testing.assert_allclose(np.hypot(d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)), 1.0, atol=1e-7)


def test_reference_bounds_error_quirk_on_cpu(oracle):
    """The reachable BoundsError of the reference (include/rcw.h, RCW_OOB_ERROR)."""
    orc = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    orc.set_state([[2, 2]], [[4.5, 4.5]], [0])
    for _ in range(19):
        orc.step([1])
    assert orc.position[0, 0] == np.float32(6.875) and orc.status[0] == 0
    orc.step([1])
    assert orc.status[0] == -5 and orc.position[0, 0] == np.float32(6.875)
    layer = _ring()
    with pytest.raises(IndexError):
        oracle.is_player_colliding(np.zeros((8, 8), bool), np.float32(7.0), np.float32(4.5), np.float32(0.125))
    assert oracle.is_player_colliding(layer, np.float32(7.0), np.float32(4.5), np.float32(0.125), oob_empty=True)
