"""Known answers derived by hand from the reference text (tests/golden/hand_derived.json), checked
against BOTH implementations:

  backend "oracle"  the CPU restatement (oracle/rcw_oracle.c) — runs without a GPU;
  backend "hip"     the product: librcw_hip through the C ABI (`-m gpu`), with no oracle in the loop.

Render and dynamics cases go through set_state / act / the getters on either backend.  The
stand-alone cast_ray and collision vectors are additionally replayed through a whole engine (a
custom direction table puts the wanted direction on the fan's centre ray; a forward move lands the
player exactly on the wanted position), so the HIP kernels meet every hand-derived vector directly.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "hand_derived.json")) as f:
    KAT = json.load(f)
COL = KAT["colours"]
CID = {"wall_dim_1": 0, "wall_dim_2": 1, "goal_dim_1": 2, "goal_dim_2": 3}
ROOM = dict(height_tile_map_tu=8, width_tile_map_tu=8)


class OracleWorld:
    """One reference world on the CPU oracle, behind the attribute names the cases use."""

    def __init__(self, oracle, num_rays, room=ROOM, T="Float32", **kw):
        if T == "Float64":
            kw["world_unit_bits"] = 64
        self.real = np.float64 if T == "Float64" else np.float32
        self.b = oracle.OracleBatch(1, num_rays=num_rays, **room, **kw)

    def set_state(self, goal, pos, d):
        self.b.set_state([goal], [pos], [d])

    def set_direction_table(self, dirs):
        self.b.set_direction_table(dirs)

    def step(self, action):
        rc = self.b.step([action])
        return rc if rc else int(self.b.status[0])     # -5: the reference would have raised BoundsError

    col_height = property(lambda s: s.b.col_height[0])
    col_colour = property(lambda s: s.b.col_colour[0])
    frame = property(lambda s: s.b.camera_view[0])
    ray_dim = property(lambda s: s.b.ray_dim[0])
    ray_stop = property(lambda s: s.b.ray_stop[0])
    ray_dist = property(lambda s: s.b.ray_dist[0])
    ray_dirs = property(lambda s: s.b.ray_dirs[0])
    position = property(lambda s: s.b.position[0])
    direction = property(lambda s: int(s.b.direction[0]))
    reward = property(lambda s: s.b.reward[0])
    done = property(lambda s: bool(s.b.done[0]))
    status = property(lambda s: int(s.b.status[0]))
    top_view = property(lambda s: s.b.top_view[0])

    def close(self):
        self.b.close()


class HipWorld:
    """The same world on the MI355X engine (C ABI via the Python host mirror)."""

    def __init__(self, rcw, num_rays, room=ROOM, T="Float32", **kw):
        self.rcw = rcw
        self._status = 0
        self.real = np.float64 if T == "Float64" else np.float32
        self.env = rcw.SingleRoomModule.SingleRoom(batch=1, num_rays=num_rays, T=T, **room, **kw)

    def set_state(self, goal, pos, d):
        self.env.set_state([goal], [pos], [d])

    def set_direction_table(self, dirs):
        self.env.set_direction_table(dirs)

    def step(self, action):
        self.rcw.act_(self.env, action)
        self._status = 0
        try:
            self.env.sync()
        except IndexError:                               # Julia: BoundsError (RCW_ERR_OUT_OF_BOUNDS)
            self._status = int(self.env.world.status[0])
            self.env.clear_error()                       # the sticky word would fail every later getter
        return self._status

    col_height = property(lambda s: s.env.columns()[0][0])
    col_colour = property(lambda s: s.env.columns()[1][0])
    frame = property(lambda s: s.env.camera_view_host()[0])
    ray_dim = property(lambda s: s.env.world.rays()[1][0])
    ray_stop = property(lambda s: s.env.world.rays()[0][0])
    ray_dist = property(lambda s: s.env.world.rays()[2][0])
    ray_dirs = property(lambda s: s.env.world.rays()[3][0])
    position = property(lambda s: s.env.world.player_position_wu[0])
    direction = property(lambda s: int(s.env.world.player_direction_au[0]))
    reward = property(lambda s: s.env.world.reward[0])
    done = property(lambda s: bool(s.env.world.done[0]))
    status = property(lambda s: s._status)
    top_view = property(lambda s: s.env.top_view_host()[0])

    def close(self):
        self.env.close()


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def make_world(request):
    made = []

    def make(num_rays, **kw):
        if request.param == "oracle":
            w = OracleWorld(request.getfixturevalue("oracle"), num_rays, **kw)
        else:
            w = HipWorld(request.getfixturevalue("rcw"), num_rays, **kw)
        made.append(w)
        return w

    yield make
    for w in made:
        w.close()


def _ring(H=8, W=8, goal=None):
    m = np.zeros((H, W), dtype=bool)
    m[0, :] = m[-1, :] = True
    m[:, 0] = m[:, -1] = True
    if goal is not None:
        m[goal[0] - 1, goal[1] - 1] = True
    return m


@pytest.mark.parametrize("T", ["Float32", "Float64"])
@pytest.mark.parametrize("case", KAT["render"], ids=lambda c: c["name"][:40])
def test_render_known_answers(make_world, case, T):
    """(also with T = Float64 world units, SR:259: the cases are far from rounding boundaries, so the answers hold)"""
    N = case["num_rays"]
    room = ROOM if "map" not in case else dict(height_tile_map_tu=case["map"][0], width_tile_map_tu=case["map"][1])
    w = make_world(N, room=room, T=T)
    w.set_state(case["goal"], case["position"], case["direction"])
    h, c, frame = w.col_height, w.col_colour, w.frame
    if "all_columns" in case:
        a = case["all_columns"]
        assert (h == a["h"]).all()
        assert (c == CID[a["colour"]]).all()
        assert (w.ray_dim == a["hit_dimension"]).all()
        stop = w.ray_stop
        if "stop_i" in a:
            assert (stop[:, 0] == a["stop_i"]).all()
        if "stop_j" in a:
            assert (stop[:, 1] == a["stop_j"]).all()
        if "ray1_stop_j" in case:
            assert stop[0, 1] == case["ray1_stop_j"]
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
def test_dynamics_known_answers(make_world, case):
    w = make_world(64)
    w.set_state(case["goal"], case["position"], case["direction"])
    for (action, count), expect in zip(case["actions"], case["after_each_block"]):
        for _ in range(count):
            assert w.step(action) == 0
        if "position" in expect:
            np.testing.assert_array_equal(w.position, np.array(expect["position"], dtype=np.float32))
        if "reward" in expect:
            assert w.reward == np.float32(expect["reward"])
        if "done" in expect:
            assert w.done == expect["done"]
        if "direction" in expect:
            assert w.direction == expect["direction"]


@pytest.mark.parametrize("case", KAT["cast_ray"], ids=lambda c: c["name"][:30])
def test_cast_ray_known_answers(oracle, case):
    """RayCaster.cast_ray (call site SR:223) as a stand-alone function of the restatement."""
    e = case["expect"]
    i, j, dim, dist = oracle.cast_ray(_ring(), *case["start"], *[np.float32(v) for v in case["dir"]])
    assert (i, j, dim) == (e["i"], e["j"], e["dim"])
    assert abs(float(dist) - e["dist"]) <= e.get("dist_tol", 0.0)


@pytest.mark.parametrize("case", KAT["cast_ray"], ids=lambda c: c["name"][:30])
def test_cast_ray_known_answers_through_the_engine(make_world, case):
    """The same vectors through cast_rays!(world): an odd fan (65 rays) has a centre ray whose LinRange
    element is 0.5*first + 0.5*last = the heading itself (SR:216-218), and a caller-supplied direction table
    (rcw_set_direction_table) makes heading 0 the wanted direction."""
    e = case["expect"]
    w = make_world(65)
    dirs = np.zeros((128, 2), dtype=np.float32)
    th = np.arange(128) * 2 * np.pi / 128
    dirs[:, 0], dirs[:, 1] = np.cos(th), np.sin(th)
    dirs[0] = np.array(case["dir"], dtype=np.float32)
    w.set_direction_table(dirs)
    goal = [7, 7] if case["start"][0] < 4 else [2, 2]          # a goal tile out of the ray's way
    w.set_state(goal, case["start"], 0)
    mid = 32                                                    # ray i = 33 of 65 (0-based 32)
    got_dir = w.ray_dirs[mid]
    np.testing.assert_allclose(got_dir, np.array(case["dir"], dtype=np.float32), rtol=0, atol=1e-7)
    stop, dim, dist = w.ray_stop[mid], w.ray_dim[mid], w.ray_dist[mid]
    assert (int(stop[0]), int(stop[1]), int(dim)) == (e["i"], e["j"], e["dim"])
    assert abs(float(dist) - e["dist"]) <= max(e.get("dist_tol", 0.0), 0.0 if got_dir[0] in (0.0, 1.0) else 1e-5)


@pytest.mark.parametrize("case", KAT["collision"], ids=lambda c: c["name"][:30])
def test_collision_known_answers(oracle, case):
    """is_player_colliding (CD:21-42) as a stand-alone function of the restatement."""
    if case["layer"] == "wall":
        layer = _ring()
    else:
        layer = np.zeros((8, 8), dtype=bool)
        layer[case["goal"][0] - 1, case["goal"][1] - 1] = True
    got = oracle.is_player_colliding(layer, np.float32(case["position"][0]), np.float32(case["position"][1]),
                                     np.float32(0.125))
    assert got == case["expect"]


def _approach(p):
    """(start position, heading) such that ONE forward move of 1/8 tests exactly position p (Float32)."""
    p = np.array(p, dtype=np.float32)
    inc = np.float32(0.125)
    for heading, sign in ((0, 1), (64, -1)):                    # dir (1, 0) / (-1, 1.2e-16)
        start = np.array([p[0] - np.float32(sign) * inc, p[1]], dtype=np.float32)
        if np.float32(start[0] + np.float32(sign) * inc) == p[0] and 1.0 <= start[0] < 7.0:
            return start, heading
    raise AssertionError(f"no exact approach to {p}")


@pytest.mark.parametrize("case", KAT["collision"], ids=lambda c: c["name"][:30])
def test_collision_known_answers_through_the_engine(make_world, case):
    """The same vectors through act!(world, MOVE_FORWARD) SR:150-176: the candidate position is the case's;
    a hit on the wall layer blocks the move, a hit on the goal layer gives reward 1 / done, a miss moves."""
    w = make_world(64)
    start, heading = _approach(case["position"])
    goal = case.get("goal", [7, 7] if case["position"][0] < 4 else [2, 2])
    w.set_state(goal, start, heading)
    assert w.step(1) == 0
    want = np.array(case["position"], dtype=np.float32)
    if case["layer"] == "wall":
        if case["expect"]:
            np.testing.assert_array_equal(w.position, start)    # SR:170-171
        else:
            np.testing.assert_array_equal(w.position, want)     # SR:174
        assert w.reward == 0 and not w.done
    else:
        if case["expect"]:
            np.testing.assert_array_equal(w.position, start)    # SR:166-168: pose unchanged
            assert w.reward == 1 and w.done
        else:
            np.testing.assert_array_equal(w.position, want)
            assert w.reward == 0 and not w.done


def test_direction_table_facts(oracle):
    """dir[0] = (1, 0) with an exact zero, the only one for nd = 128."""
    d = oracle.direction_table(128)
    assert d[0, 0] == 1.0 and d[0, 1] == 0.0
    assert np.count_nonzero(d == 0.0) == 1
    assert d[64, 0] == -1.0 and d[64, 1] == np.float32(1.2246469e-16)
    assert d[32, 0] == np.float32(6.123234e-17) and d[32, 1] == 1.0
    np.testing.assert_allclose(np.hypot(d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)), 1.0, atol=1e-7)


def test_reference_bounds_error_quirk(make_world):
    """The reachable BoundsError of the reference (include/rcw.h, RCW_OOB_ERROR), on either backend."""
    w = make_world(64)
    w.set_state([2, 2], [4.5, 4.5], 0)
    for _ in range(19):
        assert w.step(1) == 0
    assert w.position[0] == np.float32(6.875) and w.status == 0
    assert w.step(1) == -5
    assert w.status == -5 and w.position[0] == np.float32(6.875)


def test_reference_bounds_error_quirk_stand_alone(oracle):
    layer = _ring()
    with pytest.raises(IndexError):
        oracle.is_player_colliding(np.zeros((8, 8), bool), np.float32(7.0), np.float32(4.5), np.float32(0.125))
    assert oracle.is_player_colliding(layer, np.float32(7.0), np.float32(4.5), np.float32(0.125), oob_empty=True)


def test_top_view_certain_parts_by_hand(make_world):
    """The parts of the top view that do not depend on SimpleDraw's rasterisers, derived by hand from SR:342-372 and
    SR:466-480: tile (i, j) occupies pixels (i-1)pu+1..i pu in both axes, its outermost pixel ring is 0x00cccccc, the
    inside is white for walls, red for the goal, black for free tiles; the player's own pixel wu_to_pu.(position) lies
    on the first pixel of every ray line (0x00808080 unless the circle, radius floor(r pu)+1 = 5 at pu 32, covers it —
    it does not cover the centre); facing +x the central ray runs straight down the i axis to the wall face x = 7."""
    pu = 32
    w = make_world(64, render_top_view=1, pu_per_tu=pu)
    w.set_state([7, 2], [4.5, 4.5], 0)
    tv = w.top_view                           # [jp-1][ip-1]
    px = lambda ip, jp: int(tv[jp - 1, ip - 1])   # noqa: E731
    assert px(1, 1) == 0xCCCCCC and px(32, 32) == 0xCCCCCC and px(33, 1) == 0xCCCCCC      # frames
    assert px(2, 2) == 0xFFFFFF and px(16, 16) == 0xFFFFFF                                # wall tile (1,1) inside
    assert px(6 * pu + 2, 1 * pu + 2) == 0xFF0000 and px(7 * pu - 1, 2 * pu - 1) == 0xFF0000   # goal tile (7,2) inside
    assert px(6 * pu + 1, 1 * pu + 1) == 0xCCCCCC                                          # goal tile frame
    assert px(2 * pu + 2, 2 * pu + 2) == 0x000000                                          # free tile (3,3), away from the fan
    ip = jp = int(np.floor(4.5 * pu)) + 1                                                  # 145
    assert px(ip, jp) == 0x808080                                                          # start of every ray line
    assert px(ip + 5, jp) == 0xC0C0C0 and px(ip, jp + 5) == 0xC0C0C0 and px(ip - 5, jp) == 0xC0C0C0   # circle's axis points
    assert px(200, jp) == 0x808080 and px(225, jp) == 0x808080 and px(226, jp) != 0x808080
    # the whole image holds nothing but the six colours of SR:288-290, SR:364
    assert set(np.unique(tv).tolist()) <= {0x000000, 0xFFFFFF, 0xFF0000, 0xCCCCCC, 0x808080, 0xC0C0C0}
