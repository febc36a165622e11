"""GPU parity: librcw_hip (through the C ABI) vs the CPU oracle on identical inputs.

Bit-exact for every integer output (tile_map, camera_view, ray stop tiles, hit dimension,
height_line_pu, colour id, direction, done); player_position_wu within 1e-6 as
BASELINE.json's north_star states (and in fact bit-exact).
"""
import gc
import os

import numpy as np
import pytest

from helpers import CFG1, CFG2, CFG3, CFG4, CFG5, REFERENCE_DEFAULT, assert_state_equal

pytestmark = pytest.mark.gpu


def _make(rcw, oracle, batch, seed=0, **kw):
    env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=seed, **kw)
    okw = {k: v for k, v in kw.items() if k not in ("auto_reset", "T", "library")}
    if kw.get("auto_reset"):
        okw["auto_reset"] = 1
    if kw.get("T") == "Float64":
        # convert(Float64, .) of the same kwargs (SR:263-270)
        okw["world_unit_bits"] = 64
        for k in ("player_radius_wu", "position_increment_wu", "semi_field_of_view_wu", "camera_height_tile_wu"):
            if k in kw:
                okw[k + "_f64"] = float(kw[k])
    orc = oracle.OracleBatch(batch, seed=seed, **okw)
    return env, orc


def _rollout(rcw, env, orc, steps, rng, check_every=1, rays_every=0, frames=True):
    for s in range(steps):
        a = rng.integers(1, 5, env.batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
        if (s + 1) % check_every == 0 or s == steps - 1:
            assert_state_equal(env, orc, frames=frames, rays=bool(rays_every) and (s % rays_every == 0),
                               where=f"step {s}")


@pytest.mark.parametrize("cfg,batch,steps", [
    (CFG1, 1, 300),              # configs[0]: the reference's own CPU-runnable case
    (CFG2, 64, 120),
    (REFERENCE_DEFAULT, 32, 120),   # 8 x 16, 512 rays (test/runtests.jl:19)
    (CFG3, 16, 80),
    (CFG4, 32, 80),
    (CFG5, 8, 80),               # deep DDA march
])
def test_reset_and_rollout_match_oracle(rcw, oracle, cfg, batch, steps):
    env, orc = _make(rcw, oracle, batch, seed=1234, **cfg)
    assert_state_equal(env, orc, rays=True, where="after create/reset")
    rng = np.random.default_rng(7)
    _rollout(rcw, env, orc, steps, rng, check_every=1 if batch <= 8 else 10, rays_every=20)
    # masked reset with a new seed
    mask = (rng.random(batch) < 0.5).astype(np.uint8)
    mask[0] = 1
    rcw.reset_(env, mask=mask, seed=99)
    orc.reset(mask=mask, seed=99)
    assert_state_equal(env, orc, rays=True, where="after masked reset")
    _rollout(rcw, env, orc, 20, rng, check_every=5)
    env.close()


def test_set_state_all_headings(rcw, oracle):
    """Every heading from a fixed off-centre pose: exercises the whole ray table."""
    nd = 128
    env, orc = _make(rcw, oracle, nd, **CFG1)
    goal = np.tile(np.array([[3, 6]], dtype=np.int32), (nd, 1))
    pos = np.tile(np.array([[4.3125, 2.71875]], dtype=np.float32), (nd, 1))
    d = np.arange(nd, dtype=np.int32)
    env.set_state(goal, pos, d)
    orc.set_state(goal, pos, d)
    assert_state_equal(env, orc, rays=True, where="all headings")
    env.close()


def test_unpinned_switches_match_oracle(rcw, oracle):
    """The three UNPINNED choices (include/rcw.h) are switchable and stay in parity."""
    rng = np.random.default_rng(3)
    for tie in (0, 1):
        for dist in (0, 1):
            for norm in (0, 1):
                env, orc = _make(rcw, oracle, 16, seed=5, dda_tie_break=tie, dda_distance=dist,
                                 normalize_mode=norm, **CFG1)
                _rollout(rcw, env, orc, 40, rng, check_every=10, rays_every=10)
                env.close()


def test_auto_reset_matches_oracle(rcw, oracle):
    env, orc = _make(rcw, oracle, 256, seed=11, auto_reset=True, **CFG1)
    rng = np.random.default_rng(5)
    episodes_before = env.world.episode.copy()
    _rollout(rcw, env, orc, 400, rng, check_every=50, frames=False)
    np.testing.assert_array_equal(env.world.episode, orc.episode)
    assert (env.world.episode > episodes_before).any(), "no agent finished an episode in 400 random steps"
    env.close()


def test_invalid_action_mutates_nothing(rcw, oracle):
    env, orc = _make(rcw, oracle, 8, seed=2, **CFG1)
    before = env.camera_view_host().copy()
    a = np.array([1, 2, 3, 4, 1, 5, 1, 1], dtype=np.uint8)
    with pytest.raises(AssertionError):
        rcw.act_(env, a)
    assert orc.step(a) == -2
    with pytest.raises(AssertionError):
        rcw.act_(env, 0)
    assert_state_equal(env, orc, where="after rejected actions")
    np.testing.assert_array_equal(env.camera_view_host(), before)
    env.close()


def test_device_actions_and_validation(rcw, oracle):
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 32, seed=4, **CFG2)
    rng = np.random.default_rng(9)
    for s in range(30):
        a = rng.integers(1, 5, env.batch).astype(np.uint8)
        rcw.act_(env, torch.from_numpy(a).cuda())
        orc.step(a)
    assert_state_equal(env, orc, where="device actions")
    bad = rng.integers(1, 5, env.batch).astype(np.uint8)
    bad[17] = 9
    bad[3] = 0
    rcw.act_(env, torch.from_numpy(bad).cuda())   # asynchronous: surfaces at the next sync / getter
    orc.step_lenient(bad)
    with pytest.raises(AssertionError):
        env.sync()
    np.testing.assert_array_equal(env.world.status, orc.status)
    assert env.world.status[17] == -2 and env.world.status[3] == -2
    env.clear_error()
    orc.clear_status()
    assert_state_equal(env, orc, where="after device actions with two invalid entries")
    env.close()


def test_state_aliases_device_buffer(rcw, oracle):
    """RLBase.state(env) aliases camera_view (SR:576): same pointer, updated in place."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 4, seed=6, **CFG1)
    rl = rcw.RLBaseEnv(env)
    s0 = rcw.RLBase.state(rl)
    t = s0.torch()
    rl(3)
    orc.step(np.full(4, 3, dtype=np.uint8))
    s1 = rcw.RLBase.state(rl)
    assert s0.ptr == s1.ptr
    env.sync()
    np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32), orc.camera_view)
    # the same alias through DLPack (what jax / cupy would take): no copy, the engine's pointer
    d = torch.from_dlpack(s1)
    assert d.data_ptr() == s0.ptr and d.dtype == torch.uint32 and tuple(d.shape) == s0.shape
    assert torch.from_dlpack(rcw.RLBase.is_terminated(rl)).dtype == torch.bool
    rl(4)
    orc.step(np.full(4, 4, dtype=np.uint8))
    env.sync()
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), orc.camera_view)
    env.close()


def test_expand_columns_reproduces_frames(rcw, oracle):
    """The compact descriptor + expansion (receiving side of the observation gather)."""
    torch = pytest.importorskip("torch")
    import ctypes as C

    env, orc = _make(rcw, oracle, 16, seed=8, **CFG2)
    h, c = env.columns_device()
    out = torch.zeros((16, 256, 256), dtype=torch.int32, device="cuda")
    from raycastworlds_jl_amd import _capi
    _capi.check(env._lib.rcw_expand_columns(env._h, C.c_void_p(h.ptr), C.c_void_p(c.ptr), 16,
                                            C.c_void_p(out.data_ptr())))
    env.sync()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), orc.camera_view)
    env.close()


def test_rlbase_verbs_are_device_resident_and_do_not_synchronise(rcw, oracle):
    """RLBase.state / reward / is_terminated (SR:576-584) hand out the engine's own device arrays: the same object on
    every call, refreshed by every action in stream order.  The loop of the reference's test (test/runtests.jl:26-33) with
    a GPU-resident consumer issues NO host synchronisation; used as host arrays the same objects copy on demand."""
    torch = pytest.importorskip("torch")
    B = 64
    env, orc = _make(rcw, oracle, B, seed=8, out_of_bounds=1, **CFG2)
    rl = rcw.RLBaseEnv(env)
    RLBase = rcw.RLBase
    stream = torch.cuda.Stream()
    env.set_stream(stream.cuda_stream)
    rng = np.random.default_rng(6)
    acts = rng.integers(1, 5, (40, B)).astype(np.uint8)
    with torch.cuda.stream(stream):
        dev_acts = torch.from_numpy(acts).cuda()
        returns = torch.zeros(B, dtype=torch.float32, device="cuda")
        ended = torch.zeros(B, dtype=torch.int32, device="cuda")
        stream.synchronize()
        before = env.host_syncs
        for s in range(40):
            state = RLBase.state(rl)
            rl(dev_acts[s])
            r, d = RLBase.reward(rl), RLBase.is_terminated(rl)
            assert r is RLBase.reward(rl) and d is RLBase.is_terminated(rl) and state is RLBase.state(rl)
            returns += r.torch(sync=False)
            ended += d.torch(sync=False)
        assert env.host_syncs == before                                      # not one synchronisation in 40 steps
        stream.synchronize()
    want_r, want_e = np.zeros(B, np.float32), np.zeros(B, np.int32)
    for s in range(40):
        assert orc.step(acts[s]) == 0
        want_r += orc.reward; want_e += (orc.done != 0)
    np.testing.assert_array_equal(returns.cpu().numpy(), want_r)
    np.testing.assert_array_equal(ended.cpu().numpy(), want_e)
    # the same objects as host arrays: copied at the moment of use
    assert RLBase.is_terminated(rl).dtype == np.bool_ and RLBase.reward(rl).dtype == np.float32
    np.testing.assert_array_equal(np.asarray(RLBase.reward(rl)), orc.reward)
    np.testing.assert_array_equal(np.asarray(RLBase.is_terminated(rl)), orc.done != 0)
    assert ((RLBase.reward(rl) == 0) | RLBase.is_terminated(rl)).all()
    assert env.host_syncs > before
    env.close()


def test_odd_camera_heights(rcw, oracle):
    """H_cam other than 256 (height_camera_view_pu, SR:271): the moving-window kernel for 64 / 128 / 512 / 768 (chunks of
    whole columns or row blocks); rcw_fill_flat_kernel — 256-pixel chunks of the flat batch, each lane finding its own
    column — for every other height from 24 rows (84, 100, 300, 36, 24: 16-byte groups inside one column; 250, 37, 99, 257, 27,
    31, 25: groups that straddle two columns; batches whose pixel count is not a multiple of 256 or of 4: a short last chunk;
    up to twelve columns a chunk at 24 rows); the frame-per-workgroup kernel below 24 rows (20: 16-byte stores; 21, 23:
    4-byte stores); with a masked reset in between
    (the mask path of each kernel: chunks at a masked agent's border are written pixel by pixel)."""
    rng = np.random.default_rng(1)
    want = {64: "rcw_fill_window_kernel", 128: "rcw_fill_window_kernel", 512: "rcw_fill_window_kernel", 768: "rcw_fill_window_kernel",
            20: "rcw_fill_frame_kernel", 21: "rcw_fill_frame_kernel", 23: "rcw_fill_frame_kernel"}
    for hc, cfg in ((64, CFG1), (128, CFG1), (512, CFG1), (768, CFG1), (250, CFG1), (37, CFG1), (84, CFG2), (100, CFG2), (300, CFG1),
                    (99, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=33)),     # 9 x 33 x 99 pixels: not a multiple of 4
                    (257, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=65)),
                    (1000, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=31)),
                    (36, CFG1), (21, CFG1), (24, CFG2), (27, CFG1), (31, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=100)),
                    (25, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=7)), (20, CFG1), (23, CFG1),
                    (64, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=66)),
                    (128, dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=33))):
        env, orc = _make(rcw, oracle, 9, seed=3, height_camera_view_pu=hc, **cfg)
        if (hc * cfg["num_rays"]) % 256 == 0 or hc not in (64, 128):
            assert env.fill_kernel_name() == want.get(hc, "rcw_fill_flat_kernel"), (hc, env.fill_kernel_name())
        _rollout(rcw, env, orc, 12, rng, check_every=4)
        mask = np.array([1, 0, 0, 1, 1, 0, 1, 0, 1], dtype=np.uint8)
        rcw.reset_(env, mask=mask, seed=8); orc.reset(mask=mask, seed=8)
        assert_state_equal(env, orc, where=f"masked reset, H_cam {hc}")
        _rollout(rcw, env, orc, 8, rng, check_every=4)
        env.close()


def test_reference_bounds_error_quirk(rcw, oracle):
    """Walking +x in exact 1/8 steps reaches x = H-1-1/8 (not colliding: strict `<`, CD:18);
    the next forward move indexes tile H+1 -> BoundsError in the reference (CD:35).  The
    engine reports it, leaves that agent as it was and keeps stepping the others."""
    env, orc = _make(rcw, oracle, 4, seed=0, **CFG1)
    goal = np.array([[2, 2]] * 4, dtype=np.int32)
    pos = np.array([[4.5, 4.5], [4.5, 4.5], [4.5, 4.5], [2.5, 5.5]], dtype=np.float32)
    d = np.array([0, 32, 64, 0], dtype=np.int32)      # +x, +y, -x, +x
    env.set_state(goal, pos, d)
    orc.set_state(goal, pos, d)
    fwd = np.ones(4, dtype=np.uint8)
    for _ in range(19):                                # 4.5 + 19/8 = 6.875
        rcw.act_(env, fwd)
        orc.step(fwd)
    env.sync()                                         # no error yet
    assert env.world.player_position_wu[0, 0] == np.float32(6.875)
    rcw.act_(env, fwd)
    orc.step(fwd)
    with pytest.raises(IndexError):
        env.sync()
    np.testing.assert_array_equal(env.world.status, orc.status)
    assert list(env.world.status) == [-5, -5, 0, 0]
    env.clear_error()
    assert_state_equal(env, orc, where="after BoundsError")
    turn = np.full(4, 3, dtype=np.uint8)
    rcw.act_(env, turn)
    orc.step(turn)
    assert_state_equal(env, orc, where="turn after BoundsError")
    env.close()
    # RCW_OOB_TREAT_EMPTY: same walk, the move is simply blocked by the wall
    env, orc = _make(rcw, oracle, 4, seed=0, out_of_bounds=1, **CFG1)
    env.set_state(goal, pos, d)
    orc.set_state(goal, pos, d)
    for _ in range(25):
        rcw.act_(env, fwd)
        orc.step(fwd)
    env.sync()
    assert_state_equal(env, orc, where="treat-empty policy")
    assert env.world.player_position_wu[0, 0] == np.float32(6.875)
    env.close()


def test_sampler_give_up_status_bit(rcw, oracle):
    """utils.jl:34: on a map without an empty tile (3 x 3: the one interior tile is the goal) `sample_empty_position` exhausts its
    1024 H W tries, @warns and returns the occupied tile; the engine's reset does the same 9,216 draws, goes on, and records the
    warning RCW_WARN_SAMPLER_GAVE_UP in the agent's status word — no error word, no call fails.  State, status words and frames
    against the oracle at construction, after a masked reset and through steps (a start inside an obstacle: every ray ends at once)."""
    _capi = rcw.SingleRoomModule._capi
    env, orc = _make(rcw, oracle, 6, seed=4, height_tile_map_tu=3, width_tile_map_tu=3, num_rays=16, out_of_bounds=1)
    env.sync()                                                           # a warning is no error
    np.testing.assert_array_equal(env.world.status, orc.status)
    assert (env.world.status == _capi.RCW_WARN_SAMPLER_GAVE_UP).all()
    assert_state_equal(env, orc, rays=True, where="3x3 create")
    env.clear_error(); orc.clear_status()
    mask = np.array([1, 0, 1, 0, 0, 1], np.uint8)
    rcw.reset_(env, mask, seed=9); orc.reset(mask=mask, seed=9)
    env.sync()
    np.testing.assert_array_equal(env.world.status, orc.status)
    assert list(env.world.status) == [1, 0, 1, 0, 0, 1]
    _rollout(rcw, env, orc, 6, np.random.default_rng(2), rays_every=2)
    np.testing.assert_array_equal(env.world.status, orc.status)
    env.close()
    env, orc = _make(rcw, oracle, 6, seed=4, height_tile_map_tu=4, width_tile_map_tu=3, num_rays=16)   # two interior tiles: one stays empty
    assert (env.world.status == 0).all() and (orc.status == 0).all()
    env.close()


def test_non_default_parameters(rcw, oracle):
    """Odd sizes and non-default kwargs of SingleRoom(; ...) SR:258-272: non-square map, N not a
    multiple of 64, other num_directions / fov / radius / increment / camera height."""
    rng = np.random.default_rng(11)
    cases = [
        dict(height_tile_map_tu=12, width_tile_map_tu=20, num_rays=100),
        dict(height_tile_map_tu=5, width_tile_map_tu=9, num_rays=37, num_directions=36),
        dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=200, semi_field_of_view_wu=1.0,
             camera_height_tile_wu=0.5),
        dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64, player_radius_wu=0.25,
             position_increment_wu=0.0625),
        dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64, player_radius_wu=0.1,
             position_increment_wu=0.3),          # inc > radius: the BoundsError regime
        dict(height_tile_map_tu=3, width_tile_map_tu=3, num_rays=64),   # smallest room: one free tile
        dict(height_tile_map_tu=40, width_tile_map_tu=40, num_rays=96),
    ]
    for kw in cases:
        env, orc = _make(rcw, oracle, 24, seed=21, **kw)
        assert_state_equal(env, orc, rays=True, where=f"create {kw}")
        for s in range(60):
            a = rng.integers(1, 5, env.batch).astype(np.uint8)
            rcw.act_(env, a)
            assert orc.step(a) == 0
            try:
                env.sync()
            except IndexError:
                np.testing.assert_array_equal(env.world.status, orc.status)
                env.clear_error()
                orc.clear_status()
        assert_state_equal(env, orc, rays=True, where=f"rollout {kw}")
        env.close()


def test_custom_direction_table(rcw, oracle):
    """rcw_set_direction_table: a caller-supplied directions_wu (e.g. Julia's own cos/sin) replaces
    SR:65-69 and the ray table is rebuilt from it."""
    env, orc = _make(rcw, oracle, 8, seed=3, **CFG1)
    nd = 128
    th = (np.arange(nd) * 2 * np.pi / nd) + 0.01
    dirs = np.stack([np.cos(th), np.sin(th)], axis=1).astype(np.float32)
    env.set_direction_table(dirs)
    orc.set_direction_table(dirs)
    assert_state_equal(env, orc, rays=True, where="custom table")
    rng = np.random.default_rng(0)
    _rollout(rcw, env, orc, 30, rng, check_every=10, rays_every=10)
    env.close()


def test_tables_config_profile_and_timer_entry_points(rcw, oracle):
    """The exports tests/test_zz_abi_call_coverage.py found no GPU test calling (round 4): the Float32 tables
    (rcw_direction_table, rcw_ray_table: SR:65-69 and the fan of SR:214-221 per heading, against the oracle's), the caller's
    table in Float64 (rcw_set_direction_table64), rcw_batch / rcw_get_config (what the handle was made with, defaults filled in),
    and the two timing aids (rcw_profile / rcw_profile_read: HIP events around each kernel of a step; rcw_timer_start / _stop)."""
    import ctypes as C

    from raycastworlds_jl_amd import _capi

    env, orc = _make(rcw, oracle, 24, seed=17, auto_reset=True, out_of_bounds=1, num_directions=96, **CFG2)
    lib = env._lib
    # Float32 tables
    np.testing.assert_array_equal(env.world.directions_wu, orc.directions)
    tab = env.ray_table()                                                    # (nd, 5, N): dx | dy | |1/dx| | |1/dy| | dir . ray
    assert tab.dtype == np.float32 and tab.shape == (96, 5, 256)
    np.testing.assert_array_equal(tab[:, 0, :], orc.ray_table[:, :, 0])
    np.testing.assert_array_equal(tab[:, 1, :], orc.ray_table[:, :, 1])
    with np.errstate(divide="ignore"):
        np.testing.assert_array_equal(tab[:, 2, :], np.abs(np.float32(1) / tab[:, 0, :]))
        np.testing.assert_array_equal(tab[:, 3, :], np.abs(np.float32(1) / tab[:, 1, :]))
    d = env.world.directions_wu
    np.testing.assert_array_equal(tab[:, 4, :], d[:, None, 0] * tab[:, 0, :] + d[:, None, 1] * tab[:, 1, :])   # SR:404, one rounding per operation
    # what the handle was made with
    n = C.c_int32()
    env._check(lib.rcw_batch(env._h, C.byref(n)))
    assert n.value == 24
    cfg = _capi.RcwConfig()
    env._check(lib.rcw_get_config(env._h, C.byref(cfg)))
    assert (cfg.abi_version, cfg.height_tile_map_tu, cfg.width_tile_map_tu, cfg.num_rays, cfg.num_directions) == (lib.rcw_abi_version(), 8, 8, 256, 96)
    assert (cfg.height_camera_view_pu, cfg.pu_per_tu, cfg.auto_reset) == (256, 32, 1) and cfg.player_radius_wu == np.float32(1 / 8)
    # profiling: events around the cast and the fill kernel of each step
    env.profile(True)
    a = np.random.default_rng(0).integers(1, 5, 24).astype(np.uint8)
    for _ in range(5):
        rcw.act_(env, a); assert orc.step(a) == 0
    cast_ms, top_ms, fill_ms, steps = env.profile_read()
    env.profile(False)
    assert steps == 5 and 0 < cast_ms < 5 and 0 < fill_ms < 5 and top_ms == 0
    env.timer_start()                                                        # (stream time between two events: nothing of the host in between)
    for _ in range(3):
        rcw.act_(env, a)
    assert 0 < env.timer_stop() < 50
    for _ in range(3):
        assert orc.step(a) == 0
    assert_state_equal(env, orc, where="after profiled and timed steps")
    env.close()
    # the caller's direction table in Float64
    env, orc = _make(rcw, oracle, 6, seed=4, T="Float64", **CFG1)
    th = (np.arange(128) * 2 * np.pi / 128) - 0.003
    dirs = np.stack([np.cos(th), np.sin(th)], axis=1)
    env.set_direction_table(dirs); orc.set_direction_table(dirs)
    np.testing.assert_array_equal(env.world.directions_wu, dirs)
    assert_state_equal(env, orc, rays=True, where="custom Float64 table")
    _rollout(rcw, env, orc, 20, np.random.default_rng(1), check_every=10, rays_every=10)
    env.close()


def test_top_view_matches_oracle(rcw, oracle):
    """update_top_view! SR:446-483 (opt-in): tiles + grid, one line per ray, the player circle."""
    rng = np.random.default_rng(13)
    for kw in (dict(pu_per_tu=32, **CFG1), dict(pu_per_tu=32), dict(pu_per_tu=10, **CFG2),
               dict(pu_per_tu=7, height_tile_map_tu=9, width_tile_map_tu=12, num_rays=100),
               dict(pu_per_tu=32, **CFG3)):      # 512 x 512 px
        env, orc = _make(rcw, oracle, 12, seed=17, render_top_view=1, **kw)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after reset {kw}")
        for s in range(40):
            a = rng.integers(1, 5, env.batch).astype(np.uint8)
            rcw.act_(env, a)
            assert orc.step(a) == 0
            try:
                env.sync()
            except IndexError:
                env.clear_error(); orc.clear_status()
            if s % 8 == 7:
                np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"step {s} {kw}")
        mask = (rng.random(env.batch) < 0.5).astype(np.uint8)
        rcw.reset_(env, mask=mask, seed=5)
        orc.reset(mask=mask, seed=5)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after masked reset {kw}")
        assert_state_equal(env, orc, where=f"camera path unaffected {kw}")
        env.close()


def test_caller_stream_and_caller_buffer(rcw, oracle):
    """rcw_set_stream + rcw_bind_obs: the engine runs on torch's stream and renders straight into a
    torch tensor, so torch ops on that stream see the frames without any host synchronisation."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 64, seed=19, **CFG2)
    stream = torch.cuda.Stream()
    frames = torch.zeros((2, 64, 256, 256), dtype=torch.int32, device="cuda")     # two observation slots
    torch.cuda.synchronize()
    env.set_stream(stream.cuda_stream)
    rng = np.random.default_rng(4)
    sums = []
    with torch.cuda.stream(stream):
        for s in range(20):
            env.bind_obs(frames[s & 1].data_ptr())                                  # double-buffered observations
            a = rng.integers(1, 5, 64).astype(np.uint8)
            rcw.act_(env, torch.from_numpy(a).to("cuda", non_blocking=False))
            orc.step(a)
            sums.append((frames[s & 1].to(torch.int64).sum(), int(orc.camera_view.astype(np.int64).sum())))
    stream.synchronize()
    for got, want in sums:
        assert int(got) == want
    np.testing.assert_array_equal(frames[1].cpu().numpy().view(np.uint32), orc.camera_view)   # step 19 -> slot 1
    env.set_stream(None)
    env.bind_obs(None)
    rcw.act_(env, 3); orc.step(np.full(64, 3, np.uint8))
    assert_state_equal(env, orc, where="back on the handle's own stream and buffer")
    env.close()


def test_error_paths_fail_loudly(rcw):
    """Bad configs / arguments come back as errors (never a silent fallback), with a message."""
    SR = rcw.SingleRoomModule.SingleRoom
    from raycastworlds_jl_amd import _capi

    for bad in (dict(height_tile_map_tu=2), dict(num_rays=0), dict(player_radius_wu=0.5), dict(player_radius_wu=0.0),
                dict(position_increment_wu=-1.0), dict(semi_field_of_view_wu=float("nan")), dict(num_directions=0),
                dict(height_camera_view_pu=0), dict(dda_tie_break=7), dict(out_of_bounds=3)):
        with pytest.raises(ValueError):
            SR(batch=4, **bad)
    with pytest.raises(ValueError):
        SR(batch=0)
    with pytest.raises(_capi.RcwError) as ei:
        SR(batch=4, device=99)
    assert ei.value.code == _capi.RCW_ERR_NO_DEVICE
    with pytest.raises(NotImplementedError):
        SR(batch=4, T="Float16")
    env = SR(batch=4, **CFG1)
    with pytest.raises(ValueError):
        env.set_state([[1, 2]] * 4, [[4.5, 4.5]] * 4, [0] * 4)            # goal on the wall ring
    with pytest.raises(ValueError):
        env.set_state([[2, 2]] * 4, [[0.5, 4.5]] * 4, [0] * 4)            # player outside the room
    with pytest.raises(ValueError):
        env.set_state([[2, 2]] * 4, [[4.5, 4.5]] * 4, [128] * 4)          # heading out of range
    with pytest.raises(ValueError):
        env.camera_view_host(2, 5)                                         # agent range past the batch
    with pytest.raises(ValueError):
        rcw.act_(env, [1, 2, 3])                                           # wrong number of actions
    with pytest.raises(_capi.RcwError):
        env.top_view_host()                                                # not built with render_top_view
    env.sync()                                                             # the handle is still healthy
    rcw.act_(env, 1)
    env.sync()
    env.close()


def test_small_conveniences_of_the_mirror(rcw, oracle):
    """What a line census of the Python mirror (round 4) found no test executing: DeviceArray's array-like protocol (nbytes, all,
    sum, indexing, repr), the world's bool tile map (SR:22, unpacked from the BitArray chunks) and scalar properties, the
    environment as a context manager, device_name, RLBaseEnv's repr, the refusals of malformed descriptor / action / form
    arguments, host torch actions, and the 16-deep ring of held references on a stream torch does not own."""
    torch = pytest.importorskip("torch")
    SR = rcw.SingleRoomModule.SingleRoom
    with SR(batch=40, seed=5, auto_reset=True, out_of_bounds=1, **CFG1) as env:
        orc = oracle.OracleBatch(40, seed=5, auto_reset=1, out_of_bounds=1, **CFG1)
        rl = rcw.RLBaseEnv(env)
        assert "batch=40" in repr(rl) and "8x8" in repr(rl) and "gfx950" in env.device_name()
        w = env.world
        assert (w.num_rays, w.num_directions) == (64, 128) and w.player_radius_wu == np.float32(1 / 8)
        assert w.position_increment_wu == np.float32(1 / 8) and w.semi_field_of_view_wu == np.float32(2 / 3)
        # tile_map[b, o, i, j]: the wall ring, one goal tile where goal_position says (SR:54-63)
        tm = w.tile_map
        assert tm.shape == (40, 2, 8, 8) and tm.dtype == bool
        ring = np.ones((8, 8), bool); ring[1:-1, 1:-1] = False
        assert (tm[:, 0] == ring).all() and (tm[:, 1].sum(axis=(1, 2)) == 1).all()
        g = w.goal_position
        assert all(tm[b, 1, g[b, 0] - 1, g[b, 1] - 1] for b in range(40))
        # DeviceArray: a device-resident array that behaves like its host copy
        a = np.random.default_rng(0).integers(1, 5, 40).astype(np.uint8)
        rcw.act_(env, a); assert orc.step(a) == 0
        cam = env.camera_view
        assert cam.nbytes == 40 * 64 * 256 * 4 and "DeviceArray(ptr=0x" in repr(cam) and len(cam) == 40
        np.testing.assert_array_equal(cam[3], orc.camera_view[3])
        np.testing.assert_array_equal(cam[5:7, ::8], orc.camera_view[5:7, ::8])
        rew = env.reward_device()
        assert rew.sum() == orc.reward.sum() and rew.all() == orc.reward.all() and not env.done_device().any()
        # actions as a HOST torch tensor, and malformed device ones
        acts = torch.from_numpy(a).cuda()
        rcw.act_(env, torch.from_numpy(a)); assert orc.step(a) == 0
        with pytest.raises(ValueError, match="contiguous uint8"):
            rcw.act_(env, acts.to(torch.int32))
        with pytest.raises(ValueError, match="contiguous uint8"):
            rcw.act_(env, acts[:17])
        h, c = env.columns_device()
        with pytest.raises(ValueError, match="shape"):
            env.expand_columns(h.torch()[:, :10], c.torch()[:, :10])
        with pytest.raises(ValueError, match="int32 / uint8"):
            env.expand_columns(h.torch().float(), c.torch())
        with pytest.raises(ValueError, match="unknown top view form"):
            env.set_top_view_form("sideways")
        assert_state_equal(env, orc, where="after the refusals")
        # a raw stream (not torch's to keep alive): references to device actions are held in a ring of 16 events
        raw = torch.cuda.Stream()
        env.set_stream(raw.cuda_stream)
        rng = np.random.default_rng(1)
        for _ in range(60):                                                  # (more launches than the ring is deep; temporaries dropped at once)
            a = rng.integers(1, 5, 40).astype(np.uint8)
            rcw.act_(env, (torch.from_numpy(a).cuda() + 0).to(torch.uint8))
            assert orc.step(a) == 0
        assert 1 <= len(env._held) <= 16
        assert_state_equal(env, orc, where="60 steps on a raw stream")
        env.sync()
    assert env._handle is None or not env._handle.h                          # (the context manager closed it)
    from raycastworlds_jl_amd import viewer
    with pytest.raises(ValueError):
        viewer.frame_to_rgb(np.zeros((2, 3, 4), np.uint32))


def test_handles_on_concurrent_host_threads(rcw, oracle):
    """Six host threads, a handle each (different geometries, world-unit types, top-view forms — among them the first handle of
    the process that needs more than 64 KiB of LDS for its top view, i.e. the once-per-device kernel attribute), created, stepped
    and read concurrently (ctypes releases the interpreter lock inside the library): what is per process in the library — the
    error text (thread-local), the kernel attributes (a mutex) — holds, every thread's rollout equals its oracle's, and an
    error raised in one thread (an invalid action) shows in that thread only."""
    import threading

    cases = [dict(CFG1), dict(CFG2, T="Float64"), dict(CFG1, render_top_view=True, pu_per_tu=32),
             dict(height_tile_map_tu=24, width_tile_map_tu=24, num_rays=64, render_top_view=True, pu_per_tu=32),     # 768 x 768 px: 72 KiB bit plane
             dict(CFG1, height_camera_view_pu=100), dict(height_tile_map_tu=9, width_tile_map_tu=7, num_rays=33, render_top_view=True, pu_per_tu=13)]
    failures, barrier = [], threading.Barrier(len(cases))

    def worker(k, kw):
        try:
            rng = np.random.default_rng(100 + k)
            barrier.wait()
            env, orc = _make(rcw, oracle, 20 + k, seed=40 + k, auto_reset=True, out_of_bounds=1, **kw)
            for s in range(30):
                a = rng.integers(1, 5, env.batch).astype(np.uint8)
                if k == 0 and s == 10:                                       # this thread's error, and its text, stay in this thread
                    bad = a.copy(); bad[3] = 9
                    with pytest.raises(AssertionError, match="Invalid action: 9"):
                        rcw.act_(env, bad)
                    assert orc.step(bad) == -2
                rcw.act_(env, a)
                assert orc.step(a) == 0
                if s % 10 == 9:
                    assert_state_equal(env, orc, where=f"thread {k} step {s}")
                    if kw.get("render_top_view"):
                        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
            env.close()
        except BaseException as e:                                           # noqa: BLE001 (reported by the main thread)
            failures.append((k, repr(e)[:600]))
            try:
                barrier.abort()
            except Exception:                                                # noqa: BLE001
                pass

    threads = [threading.Thread(target=worker, args=(k, kw)) for k, kw in enumerate(cases)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not failures, failures
    assert not any(t.is_alive() for t in threads)


def test_c_level_validation(rcw):
    """The refusals of the C ABI itself — what a C or Julia caller meets; the Python mirror checks most of these on its own
    side first, so its tests never reached them (a census of the error returns taken, round 4): every rule of
    rcw_config / rcw_create, an action outside 1..4 in rcw_step, misaligned buffers, the wrong world-unit entry point, the
    top view on a handle made without it, a second communicator — each with its code and a message, none disturbing the handle."""
    import ctypes as C

    from raycastworlds_jl_amd import _capi

    lib = _capi.load()
    INV, UNS, ACT = _capi.RCW_ERR_INVALID_ARGUMENT, _capi.RCW_ERR_UNSUPPORTED, _capi.RCW_ERR_INVALID_ACTION

    def create(batch=4, device=0, **kw):
        cfg = _capi.RcwConfig()
        assert lib.rcw_config_default(C.byref(cfg)) == 0
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.rcw_create(C.byref(cfg), batch, device, 1, C.byref(h))
        if rc == 0:
            return h
        assert not h.value and _capi.last_error(lib), kw
        return rc

    assert lib.rcw_config_default(None) == INV
    assert lib.rcw_create(None, 4, 0, 1, None) == INV
    nan, inf = float("nan"), float("inf")
    for want, kw in ((INV, dict(abi_version=3)), (INV, dict(height_tile_map_tu=2)), (INV, dict(width_tile_map_tu=2)),
                     (UNS, dict(height_tile_map_tu=300, width_tile_map_tu=300)), (INV, dict(num_directions=0)), (INV, dict(num_rays=0)),
                     (INV, dict(height_camera_view_pu=0)), (INV, dict(num_rays=(1 << 24) + 1, num_directions=1)),
                     (UNS, dict(num_directions=4096, num_rays=8192)), (UNS, dict(height_camera_view_pu=(1 << 20) + 1)),
                     (INV, dict(reward_type=7)), (INV, dict(goal_reward=nan)), (INV, dict(goal_reward_f64=inf)),
                     (INV, dict(reward_type=_capi.RCW_REWARD_INT32, goal_reward_f64=0.5)), (INV, dict(world_unit_bits=16)),
                     (INV, dict(world_unit_bits=64, player_radius_wu_f64=0.5)), (INV, dict(world_unit_bits=64, position_increment_wu_f64=nan)),
                     (INV, dict(player_radius_wu=0.5)), (INV, dict(player_radius_wu=0.0)), (INV, dict(position_increment_wu=-1.0)),
                     (INV, dict(semi_field_of_view_wu=inf)), (INV, dict(render_top_view=1, pu_per_tu=0)), (INV, dict(render_top_view=1, pu_per_tu=5000)),
                     (INV, dict(camera_height_tile_wu=0.0)), (INV, dict(dda_tie_break=2)), (INV, dict(dda_distance=-1)),
                     (INV, dict(normalize_mode=2)), (INV, dict(out_of_bounds=2))):
        assert create(**kw) == want, (kw, _capi.last_error(lib))
    assert create(batch=0) == INV and create(device=99) == _capi.RCW_ERR_NO_DEVICE and create(device=-1) == _capi.RCW_ERR_NO_DEVICE
    # a live Float32 handle without the top view
    h = create(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    assert not isinstance(h, int)
    acts = (C.c_uint8 * 4)(1, 2, 9, 4)
    assert lib.rcw_step(h, acts) == ACT and "Invalid action: 9 (agent 2)" in _capi.last_error(lib)   # @assert SR:140
    assert lib.rcw_step(h, None) == INV
    buf = C.c_void_p()
    assert lib.rcw_device_malloc(h, 4 * 64 * 256 * 4 + 64, C.byref(buf)) == 0
    assert lib.rcw_bind_obs(h, C.c_void_p(buf.value + 4)) == INV and "16-byte aligned" in _capi.last_error(lib)
    hh, cc = C.c_void_p(), C.c_void_p()
    assert lib.rcw_columns_device_ptr(h, C.byref(hh), C.byref(cc)) == 0
    assert lib.rcw_expand_columns(h, hh, cc, 4, C.c_void_p(buf.value + 8)) == INV and "16-byte aligned" in _capi.last_error(lib)
    assert lib.rcw_expand_columns(h, hh, cc, 0, buf) == INV
    out = (C.c_double * 8)()
    assert lib.rcw_position64(h, out) == UNS and "Float32" in _capi.last_error(lib)                      # the wrong world-unit entry point
    assert lib.rcw_set_state64(h, None, None, None, None) == UNS and lib.rcw_rays64(h, 0, 1, None, None, None, None) == UNS
    form = C.c_int32(-1)
    assert lib.rcw_top_view_form(h, C.byref(form)) == 0 and form.value == 0                             # "none"
    assert lib.rcw_set_top_view_form(h, _capi.RCW_TOP_VIEW_TWO_KERNELS, 0) == UNS and "render_top_view = 0" in _capi.last_error(lib)
    img = (C.c_uint32 * 16)()
    assert lib.rcw_top_view_copy(h, img, 0, 1) == UNS and lib.rcw_update_top_view(h) == UNS
    tv = C.c_void_p()
    assert lib.rcw_top_view_device_ptr(h, C.byref(tv)) == UNS
    assert lib.rcw_gather_columns(h, buf, buf) == INV and "rcw_comm_init first" in _capi.last_error(lib)
    assert lib.rcw_gather_observations(h, 0, buf) == INV
    assert lib.rcw_comm_init(h, None, 0, 1) == INV and lib.rcw_comm_init(h, buf, 1, 1) == INV and lib.rcw_comm_init(h, buf, 0, 0) == INV
    assert lib.rcw_comm_unique_id(None) == INV
    # none of it disturbed the handle
    acts = (C.c_uint8 * 4)(1, 3, 2, 4)
    assert lib.rcw_step(h, acts) == 0 and lib.rcw_sync(h) == 0
    assert lib.rcw_device_free(h, buf) == 0 and lib.rcw_destroy(h) == 0
    # a handle with the top view: the form's own argument checks
    h = create(render_top_view=1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    assert not isinstance(h, int)
    assert lib.rcw_set_top_view_form(h, 9, 0) == INV and lib.rcw_set_top_view_form(h, 0, 9) == INV and lib.rcw_set_top_view_form(h, 0, -1) == INV
    assert lib.rcw_set_top_view_form(h, _capi.RCW_TOP_VIEW_ONE_KERNEL, 0) == 0 and lib.rcw_update_top_view(h) == 0 and lib.rcw_sync(h) == 0
    assert lib.rcw_destroy(h) == 0
    # a Float64 handle refuses the Float32 entry points
    h = create(world_unit_bits=64, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    assert not isinstance(h, int)
    out32 = (C.c_float * 8)()
    assert lib.rcw_position(h, out32) == UNS and "Float64" in _capi.last_error(lib)
    assert lib.rcw_set_direction_table(h, out32) == UNS and lib.rcw_rays(h, 0, 1, None, None, None, None) == UNS
    assert lib.rcw_destroy(h) == 0


def test_live_handle_hostile_arguments(rcw, oracle):
    """Every entry point, called on a LIVE handle with each of its pointer arguments NULL and its integers at 0, -1 and
    2^31 - 1: an error code (or, where NULL / 0 has a meaning — no mask, own stream, own buffer, automatic form — success),
    never a fault, never a kernel launched on a NULL pointer (read off rcw_api.hip: each pointer is checked before use);
    afterwards the handle still steps bit-exactly."""
    import ctypes as C

    from raycastworlds_jl_amd import _capi

    env, orc = _make(rcw, oracle, 12, seed=8, out_of_bounds=1, render_top_view=True, pu_per_tu=16, **CFG1)
    lib, h = env._lib, env._h
    calls = 0
    for name, sig in _capi.SIGNATURES.items():
        if not sig or sig[0] is not C.c_void_p or name in ("rcw_destroy", "rcw_comm_unique_id"):
            continue
        for ints in (0, -1, 2**31 - 1):
            args = [h]
            for t in sig[1:]:
                if t in (C.c_float, C.c_double):
                    args.append(0.0)
                elif t is C.c_void_p or t is C.c_char_p or hasattr(t, "contents"):
                    args.append(None)
                elif t is C.c_uint64:
                    args.append(t(ints & 0xFFFFFFFF))
                else:
                    args.append(t(ints))
            rc = getattr(lib, name)(*args)
            assert isinstance(rc, int) and -8 <= rc <= 0, (name, ints, rc)
            if rc < 0:
                assert _capi.last_error(lib), name
            calls += 1
            if len(sig) == 1 or not any(t not in (C.c_void_p, C.c_char_p) and not hasattr(t, "contents") and t not in (C.c_float, C.c_double) for t in sig[1:]):
                break                                                        # (no integer argument to vary)
    assert calls >= 70
    # the handle is as usable as before: same stream rule, own buffer, automatic form; bring both sides to one state and step
    env.clear_error()
    env.set_top_view_form(None)
    rcw.reset_(env, seed=77)                                                 # (the all-NULL rcw_reset calls above were resets too: the oracle takes the state over)
    orc.set_state(env.world.goal_position, env.world.player_position_wu, env.world.player_direction_au)
    _rollout(rcw, env, orc, 12, np.random.default_rng(4), check_every=4)
    np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
    env.close()


def test_float64_world_units(rcw, oracle):
    """T = Float64 (SingleRoom(; T = Float64) SR:259): every operation of the path in Float64, compared
    with the oracle compiled for T = Float64 (positions, distances, tables as doubles)."""
    rng = np.random.default_rng(64)
    cases = [
        dict(T="Float64", **CFG1),
        dict(T="Float64", **CFG2),
        dict(T="Float64"),                                     # reference default 8 x 16, 512 rays
        dict(T="Float64", semi_field_of_view_wu=0.9, player_radius_wu=0.2, position_increment_wu=0.07,
             height_tile_map_tu=9, width_tile_map_tu=11, num_rays=77, render_top_view=True, pu_per_tu=12),
        dict(T="Float64", auto_reset=True, **CFG1),
    ]
    for kw in cases:
        env, orc = _make(rcw, oracle, 16, seed=33, **kw)
        assert env.world.player_position_wu.dtype == np.float64
        np.testing.assert_array_equal(env.world.directions_wu, orc.directions)
        tab = env.ray_table()                                  # (nd, 5, N) float64
        np.testing.assert_array_equal(tab[:, 0, :], orc.ray_table[:, :, 0])
        np.testing.assert_array_equal(tab[:, 1, :], orc.ray_table[:, :, 1])
        assert_state_equal(env, orc, rays=True, where=f"create {kw}")
        for s in range(80):
            a = rng.integers(1, 5, env.batch).astype(np.uint8)
            rcw.act_(env, a)
            assert orc.step(a) == 0
            try:
                env.sync()
            except IndexError:
                np.testing.assert_array_equal(env.world.status, orc.status)
                env.clear_error(); orc.clear_status()
            if s % 20 == 19:
                assert_state_equal(env, orc, rays=True, where=f"step {s} {kw}")
                if kw.get("render_top_view"):
                    np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        # all headings from an off-centre pose
        if env.cfg.num_directions == 128 and env.batch == 16:
            goal = np.tile(np.array([[3, 6]], dtype=np.int32), (16, 1))
            pos = np.tile(np.array([[4.3125, 2.71875]], dtype=np.float64), (16, 1))
            for d0 in range(0, 128, 16):
                d = np.arange(d0, d0 + 16, dtype=np.int32)
                env.set_state(goal, pos, d)
                orc.set_state(goal, pos, d)
                assert_state_equal(env, orc, rays=True, where=f"headings {d0}.. {kw}")
        env.close()
    # the Float32 entry points refuse a Float64 handle (and say which to use)
    env = rcw.SingleRoomModule.SingleRoom(batch=2, T="Float64", **CFG1)
    import ctypes as C
    from raycastworlds_jl_amd import _capi
    buf = np.zeros((2, 2), np.float32)
    rc = env._lib.rcw_position(env._h, buf.ctypes.data_as(C.c_void_p))
    assert rc == _capi.RCW_ERR_UNSUPPORTED and "Float64" in _capi.last_error()
    env.close()


def test_extreme_shapes(rcw, oracle):
    """The largest shapes the LDS staging admits: 4096 view columns, a 200 x 200 tile map (40 KB of
    tile bytes in LDS, rays up to 400 steps long), a 1024-pixel-high camera view."""
    rng = np.random.default_rng(21)
    for kw, batch, steps in ((dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=4096), 4, 12),
                             (dict(height_tile_map_tu=200, width_tile_map_tu=200, num_rays=128), 6, 12),
                             (dict(height_camera_view_pu=1024, **CFG1), 6, 12)):
        env, orc = _make(rcw, oracle, batch, seed=2, **kw)
        assert_state_equal(env, orc, rays=True, where=f"create {kw}")
        _rollout(rcw, env, orc, steps, rng, check_every=4, rays_every=4)
        env.close()
    with pytest.raises(rcw.SingleRoomModule._capi.RcwError):
        rcw.SingleRoomModule.SingleRoom(batch=1, height_tile_map_tu=300, width_tile_map_tu=300)   # > 65536 tiles


def test_device_actions_are_ordered_behind_their_producer(rcw, oracle):
    """Actions computed by torch kernels on torch's stream, engine on its own stream: act_ makes the
    engine wait for the producer (no host synchronisation in between)."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 4096, seed=23, out_of_bounds=1, **CFG1)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    big = torch.randn(4096, 4096, device="cuda")
    for s in range(12):
        # a deliberately slow producer: a large matmul feeds the argmax that becomes the actions
        logits = (big @ torch.randn(4096, 4, device="cuda", generator=g))
        actions = (logits.argmax(dim=1) + 1).to(torch.uint8)
        rcw.act_(env, actions)                 # no synchronize() between producer and engine
        orc.step(actions.cpu().numpy())
    assert_state_equal(env, orc, frames=False, where="ordered behind the producer")
    env.close()


def test_reward_types(rcw, oracle):
    """R of SingleRoom(; R = ...) SR:266: world.reward / goal_reward = one(R) (SR:81-82) in Float64, Int32 and
    Int64 as well as the default Float32 — element type, values and the device alias."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(8)
    for name, dt, rt in (("Float32", np.float32, 0), ("Float64", np.float64, 1), ("Int32", np.int32, 2),
                         ("Int64", np.int64, 3)):
        env = rcw.SingleRoomModule.SingleRoom(batch=512, seed=41, R=name, auto_reset=True, out_of_bounds=1, **CFG1)
        orc = oracle.OracleBatch(512, seed=41, reward_type=rt, auto_reset=1, out_of_bounds=1, **CFG1)
        assert env.world.goal_reward == dt(1) and type(env.world.goal_reward) is dt
        total = 0
        for s in range(300):
            a = rng.integers(1, 5, 512).astype(np.uint8)
            rcw.act_(env, a)
            orc.step(a)
            r = env.world.reward
            assert r.dtype == dt and orc.reward.dtype == dt
            np.testing.assert_array_equal(r, orc.reward, err_msg=f"R = {name}, step {s}")
            # test/runtests.jl:30-34: non-zero exactly on terminal steps, and then exactly one(R)
            np.testing.assert_array_equal(r != 0, env.world.done)
            assert set(np.unique(r).tolist()) <= {0, 1}
            total += int((r != 0).sum())
        assert total > 0, "no agent reached the goal in 300 steps"
        alias = env.reward_device().torch()
        assert alias.dtype == {np.float32: torch.float32, np.float64: torch.float64, np.int32: torch.int32,
                               np.int64: torch.int64}[dt]
        np.testing.assert_array_equal(alias.cpu().numpy(), env.world.reward)
        assert_state_equal(env, orc, frames=False, where=f"R = {name}")
        if rt != 0:                                            # the Float32 getter refuses the other types, loudly
            import ctypes as C
            from raycastworlds_jl_amd import _capi
            buf = np.zeros(512, np.float32)
            assert env._lib.rcw_reward(env._h, buf.ctypes.data_as(C.c_void_p)) == _capi.RCW_ERR_UNSUPPORTED
        env.close()
    with pytest.raises(NotImplementedError):
        rcw.SingleRoomModule.SingleRoom(batch=2, R="Float16", **CFG1)


def test_frame_dump_of_engine_frames(rcw, oracle, tmp_path):
    """f4 (headless stand-in for play! / copy_image_to_frame_buffer! utils.jl:64-73): the PPM written from an
    ENGINE frame equals, byte for byte, the PPM written from the oracle's frame of the same state."""
    env, orc = _make(rcw, oracle, 6, seed=77, **CFG2)
    rng = np.random.default_rng(5)
    for s in range(25):
        a = rng.integers(1, 5, 6).astype(np.uint8)
        rcw.act_(env, a)
        orc.step(a)
    for agent in range(6):
        pe, po = tmp_path / f"engine_{agent}.ppm", tmp_path / f"oracle_{agent}.ppm"
        rcw.save_agent_ppm(env, agent, str(pe))
        rcw.save_ppm(orc.camera_view[agent], str(po))
        data = pe.read_bytes()
        assert data == po.read_bytes()
        assert data.startswith(b"P6\n256 256\n255\n") and len(data) == 15 + 256 * 256 * 3
        # the blit's transpose: image row r, column k  <-  camera_view[k][r]  (utils.jl:68-70)
        rgb = np.frombuffer(data[15:], dtype=np.uint8).reshape(256, 256, 3)
        frame = env.camera_view_host(agent, 1)[0]
        for (r, k) in ((0, 0), (128, 17), (255, 255), (40, 200)):
            px = int(frame[k, r])
            assert tuple(rgb[r, k]) == ((px >> 16) & 255, (px >> 8) & 255, px & 255)
    env.close()


def test_device_actions_may_be_dropped_right_after_act(rcw, oracle):
    """The engine runs on its own stream; a temporary action tensor handed to act_ and dropped at once must
    not be recycled by torch's caching allocator before the cast kernel has read it (`Tensor.record_stream` on the
    engine's stream in act_).
    No host synchronisation anywhere in the loop; each step allocates same-sized tensors that would reuse
    the freed block immediately."""
    torch = pytest.importorskip("torch")
    B = 4096
    env, orc = _make(rcw, oracle, B, seed=29, out_of_bounds=1, **CFG1)
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    all_actions = torch.randint(1, 5, (40, B), dtype=torch.uint8, device="cuda", generator=g)
    host = all_actions.cpu().numpy()
    big = torch.randn(2048, 2048, device="cuda")
    torch.cuda.synchronize()
    for s in range(40):
        big = big @ big * 1e-3                       # keeps torch's stream busy so it runs BEHIND the host
        rcw.act_(env, all_actions[s].clone())        # a temporary: freed as soon as act_ returns
        junk = torch.full((B,), 7, dtype=torch.uint8, device="cuda")   # same size: takes the freed block if allowed to
        del junk
    for s in range(40):
        orc.step(host[s])
    env.sync()
    assert (env.world.status == 0).all(), "an agent saw a corrupted (invalid) action"
    assert_state_equal(env, orc, frames=False, where="temporaries dropped right after act_")
    env.close()


def test_render_entry_points(rcw, oracle):
    """cast_rays! / update_camera_view! / update_top_view! as separate calls (RayCastWorlds.jl:9-14):
    update_camera_view! refills the frames from the stored ray results without casting, as the reference's does."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 8, seed=13, render_top_view=1, **CFG1)
    rng = np.random.default_rng(2)
    for s in range(10):
        a = rng.integers(1, 5, 8).astype(np.uint8)
        rcw.act_(env, a); orc.step(a)
    want, want_top = env.camera_view_host().copy(), env.top_view_host().copy()
    cam, top = env.camera_view.torch(), env.top_view.torch()
    cam.view(torch.int32).zero_(); top.view(torch.int32).zero_()
    torch.cuda.synchronize()
    assert not env.camera_view_host().any()
    rcw.update_camera_view_(env)
    np.testing.assert_array_equal(env.camera_view_host(), want)
    assert not env.top_view_host().any()                         # update_camera_view! leaves the top view alone
    rcw.update_top_view_(env)
    np.testing.assert_array_equal(env.top_view_host(), want_top)
    np.testing.assert_array_equal(want_top, orc.top_view)
    stop, dim, dist, dirs = rcw.cast_rays_(env)
    np.testing.assert_array_equal(stop, orc.ray_stop)
    np.testing.assert_array_equal(dist.view(np.uint32), orc.ray_dist.view(np.uint32))
    assert_state_equal(env, orc, where="after the separate render calls")
    env.close()


def test_top_view_in_place_fallback(rcw, oracle):
    """Images whose bit plane does not fit in LDS (here 1600 x 1280 px) take the in-place kernel; a 1024 x 1024
    image is the largest write-once case (128 KiB of LDS, raised limit)."""
    rng = np.random.default_rng(17)
    for kw, batch in ((dict(pu_per_tu=32, height_tile_map_tu=50, width_tile_map_tu=40, num_rays=128), 3),
                      (dict(pu_per_tu=32, **CFG5), 3)):
        env, orc = _make(rcw, oracle, batch, seed=3, render_top_view=1, **kw)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after reset {kw}")
        for s in range(6):
            a = rng.integers(1, 5, batch).astype(np.uint8)
            rcw.act_(env, a); orc.step(a)
        try:
            env.sync()
        except IndexError:
            env.clear_error(); orc.clear_status()
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after steps {kw}")
        env.close()


# geometries of the top view: (kwargs, batch).  The first ten are the unit kernels' (whole tiles in runs of 256 / 128 / 64 /
# 32 rows); the rest only the flat store kernel takes in the two-kernel form (rcw_top_store_flat_kernel: any pixel scale
# from 9, image height a multiple of 4): pixel scales that do not divide the four-pixel lane groups (13, 10 with an odd
# tile count...), images that are not a whole number of 1 KiB chunks (chunks straddle agents), circles wider than 32 rows.
TOP_GEOMETRIES = (
    (dict(pu_per_tu=32, **CFG2), 300), (dict(pu_per_tu=16, height_tile_map_tu=16, width_tile_map_tu=9, num_rays=100), 21),
    (dict(pu_per_tu=64, height_tile_map_tu=12, width_tile_map_tu=5, num_rays=64), 7),
    (dict(pu_per_tu=8, height_tile_map_tu=32, width_tile_map_tu=20, num_rays=33, player_radius_wu=0.3, position_increment_wu=0.2), 9),
    # heights of 128 m, 64 m and 32 m rows: two / four / eight units (runs of rows of one column) to a 1 KiB chunk
    (dict(pu_per_tu=32, height_tile_map_tu=12, width_tile_map_tu=7, num_rays=128), 37),     # 384 rows
    (dict(pu_per_tu=16, **CFG2), 130),                                                     # 128 rows
    (dict(pu_per_tu=32, height_tile_map_tu=10, width_tile_map_tu=9, num_rays=200), 11),     # 320 rows
    (dict(pu_per_tu=8, height_tile_map_tu=8, width_tile_map_tu=5, num_rays=64, player_radius_wu=0.3, position_increment_wu=0.2), 260),   # 64 rows
    (dict(pu_per_tu=32, height_tile_map_tu=9, width_tile_map_tu=11, num_rays=150), 23),     # 288 rows: eight units of 32 (now the flat kernel)
    (dict(pu_per_tu=16, height_tile_map_tu=10, width_tile_map_tu=6, num_rays=90), 50),      # 160 rows
    # ---- the flat kernel's own
    (dict(pu_per_tu=13, **CFG2), 300),                                                     # 104 x 104 px: 42.25 chunks an image, four pixels straddle tiles
    (dict(pu_per_tu=10, **CFG2), 130),                                                     # 80 rows: a chunk touches five columns
    (dict(pu_per_tu=12, height_tile_map_tu=16, width_tile_map_tu=8, num_rays=96), 40),      # 192 rows of 12-pixel tiles
    (dict(pu_per_tu=20, height_tile_map_tu=9, width_tile_map_tu=7, num_rays=70), 33),       # 180 x 140 px
    (dict(pu_per_tu=24, height_tile_map_tu=8, width_tile_map_tu=16, num_rays=512), 19),     # the reference default map at 24 px a tile
    (dict(pu_per_tu=32, player_radius_wu=0.49, position_increment_wu=0.1, **CFG2), 70),     # a circle of 33 rows
    (dict(pu_per_tu=100, height_tile_map_tu=5, width_tile_map_tu=4, num_rays=40, player_radius_wu=0.45, position_increment_wu=0.3), 5),   # 500 x 400 px, circle of 93 rows
    (dict(pu_per_tu=9, height_tile_map_tu=8, width_tile_map_tu=3, num_rays=20), 77),        # 72 x 27 px: the smallest tiles, seven columns a chunk
    (dict(pu_per_tu=11, height_tile_map_tu=4, width_tile_map_tu=9, num_rays=50), 65),       # 44 rows: the shortest image
    (dict(pu_per_tu=17, height_tile_map_tu=32, width_tile_map_tu=6, num_rays=64, T="Float64"), 6),   # 544 rows, Float64 world units
)


@pytest.mark.parametrize("form", ["two-kernels", "one-kernel", "in-place"])
def test_top_view_forms_write_the_same_pixels(rcw, oracle, form):
    """The three kernel forms of update_top_view! (rcw.h: rcw_top_view_form) against the oracle on the same states:
    the two-kernel form (draw kernel beside the camera fill, moving-window store kernel) is what an eligible geometry
    takes from 256 MiB of top view a step (rcw_set_top_view_form takes it at these small batches too; at full size:
    test_gpu_full_size.py) and asks for the other two.  300 agents x 256 columns of 256 px = 76,800 chunks: more
    than one sweep of the store kernel's window (65,536), so its last group is a partial one; the masked reset
    exercises its skipped chunks; 16 px tiles put two tile rows into one lane group."""
    rng = np.random.default_rng(29)
    for kw, batch in TOP_GEOMETRIES:
        env, orc = _make(rcw, oracle, batch, seed=23, render_top_view=1, **kw)
        env.set_top_view_form(form, runs=3)   # (two-kernel form: the batch in three runs of agents, store of one beside the drawing of the next)
        assert env.top_view_form() == form, kw
        rcw.reset_(env, seed=23); orc.reset(seed=23)                         # (rendered in the chosen form)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after reset {kw}")
        for s in range(12):
            a = rng.integers(1, 5, batch).astype(np.uint8)
            rcw.act_(env, a); orc.step(a)
        try:
            env.sync()
        except IndexError:
            env.clear_error(); orc.clear_status()
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after steps {kw}")
        mask = (rng.random(batch) < 0.4).astype(np.uint8)
        rcw.reset_(env, mask=mask, seed=8); orc.reset(mask=mask, seed=8)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after masked reset {kw}")
        rcw.update_top_view_(env)                                            # the stand-alone call
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"after update_top_view! {kw}")
        assert_state_equal(env, orc, where=f"camera path unaffected {kw}")
        env.close()


def test_masked_render_right_after_a_change_of_form(rcw, oracle):
    """Found by tools/api_fuzz.py: rcw_set_top_view_form allocates the two-kernel form's line planes; the re-render that
    follows may be the one-kernel form (12-pixel tiles: stand-alone calls take it), which writes no planes; a MASKED render
    next draws the masked agents' planes only, and the flat store kernel ORs in the first words of the unmasked
    neighbour's region for the chunk the two share — uninitialised memory, unless the planes start out zero."""
    import ctypes as C

    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    kw = dict(height_tile_map_tu=5, width_tile_map_tu=5, num_rays=7, height_camera_view_pu=512, pu_per_tu=12)
    for trial in range(4):
        env, orc = _make(rcw, oracle, 64, seed=100 + trial, render_top_view=1, **kw)
        assert env.top_view_form() == "one-kernel"
        # leave all-ones blocks of the planes' size in the device allocator, for the planes to land on
        blocks = []
        for _ in range(8):
            p = C.c_void_p()
            env._check(env._lib.rcw_device_malloc(env._h, 64 * 128 * 4 + 64, C.byref(p)))
            blocks.append(p)
            alias = rcw.SingleRoomModule.DeviceArray(p.value, (64 * 128 + 16,), np.int32, env, env._sync).torch(sync=False)
            alias.fill_(-1)
        torch.cuda.synchronize()
        for p in blocks:
            env._check(env._lib.rcw_device_free(env._h, p))
        env.set_top_view_form("two-kernels", runs=2)
        rcw.update_top_view_(env)                                            # (one-kernel form: 12-pixel tiles, stand-alone)
        mask = (rng.random(64) < 0.6).astype(np.uint8)
        rcw.reset_(env, mask=mask, seed=7 + trial); orc.reset(mask=mask, seed=7 + trial)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"trial {trial}")
        assert_state_equal(env, orc, where=f"masked reset after the change of form, trial {trial}")
        env.close()


def test_stand_alone_top_view_takes_the_measured_form(rcw, oracle):
    """`update_top_view!(env)` ALONE (nothing runs beside the drawing) takes draw -> store back to back where
    that was measured faster than the one-kernel form (profiles/r05_top_view_shapes.txt): every image of the two-kernel form's
    geometries from 256 x 256 px, every pixel scale that is no multiple of 4, every tile below 16 px.  The one-kernel form
    (rcw_top_view_kernel) keeps exactly: images below 256 x 256 px at 16, 20, 24, 28 ... px a tile, and the geometries the two-kernel
    form cannot take (tiles below 8 px off the unit kernels' grid, image heights that are no multiple of 4, batches the side stream does
    not pay for).  The list is asserted here, and for every case the stand-alone call's pixels against the oracle."""
    cases = (   # kwargs, batch, form inside a step, form of the stand-alone call
        (dict(pu_per_tu=32, **CFG2), 64, "two-kernels", "two-kernels"),                                            # 256 x 256 px
        (dict(pu_per_tu=32), 8, "two-kernels", "two-kernels"),                                                     # the reference default: 256 x 512 px, 512 rays
        (dict(pu_per_tu=32, **CFG4), 8, "two-kernels", "two-kernels"),                                             # 512 x 512
        (dict(pu_per_tu=32, height_tile_map_tu=24, width_tile_map_tu=24, num_rays=256), 114, "two-kernels", "two-kernels"),  # 768 x 768 (256 MiB: planes beyond a
        (dict(pu_per_tu=32, **CFG5), 64, "two-kernels", "two-kernels"),                                            #  256-thread workgroup take the side stream) / 1024 x 1024, 1024 rays
        (dict(pu_per_tu=32, **CFG5), 8, "one-kernel", "one-kernel"),                                               # ... and below 256 MiB of them: no side stream, no planes in HBM
        (dict(pu_per_tu=8, height_tile_map_tu=32, width_tile_map_tu=32, num_rays=256), 16, "two-kernels", "two-kernels"),   # 8-px tiles, 256 x 256
        (dict(pu_per_tu=10, **CFG2), 64, "two-kernels", "two-kernels"),                                            # 10, 13: no multiple of 4
        (dict(pu_per_tu=13, **CFG2), 64, "two-kernels", "two-kernels"),
        (dict(pu_per_tu=12, **CFG2), 64, "two-kernels", "two-kernels"),                                            # tiles below 16 px
        (dict(pu_per_tu=16, **CFG2), 64, "two-kernels", "one-kernel"),                                             # 128 x 128 px ... 224 x 224: the one-kernel form's
        (dict(pu_per_tu=20, **CFG2), 64, "two-kernels", "one-kernel"),
        (dict(pu_per_tu=24, **CFG2), 64, "two-kernels", "one-kernel"),
        (dict(pu_per_tu=28, **CFG2), 64, "two-kernels", "one-kernel"),
        (dict(pu_per_tu=24, height_tile_map_tu=12, width_tile_map_tu=12, num_rays=128), 16, "two-kernels", "two-kernels"),   # 288 x 288 at 24 px
        (dict(pu_per_tu=13, height_tile_map_tu=9, width_tile_map_tu=9, num_rays=64), 16, "one-kernel", "one-kernel"),       # 117 rows: no multiple of 4
        (dict(pu_per_tu=6, height_tile_map_tu=10, width_tile_map_tu=9, num_rays=64), 16, "one-kernel", "one-kernel"),       # 6-px tiles, 60 rows
        (dict(pu_per_tu=32, height_camera_view_pu=128, **CFG2), 16, "one-kernel", "one-kernel"),                   # 4 MiB of top view beside a 128-row camera view: no side stream
        (dict(pu_per_tu=32, height_tile_map_tu=50, width_tile_map_tu=40, num_rays=128), 2, "in-place", "in-place"),         # bit plane beyond LDS
    )
    import torch

    rng = np.random.default_rng(8)
    for kw, batch, in_step, alone in cases:
        env, orc = _make(rcw, oracle, batch, seed=6, render_top_view=1, out_of_bounds=1, **kw)
        assert (env.top_view_form(), env.update_top_view_form()) == (in_step, alone), (kw, env.top_view_form(), env.update_top_view_form())
        for _ in range(3):
            a = rng.integers(1, 5, batch).astype(np.uint8)
            rcw.act_(env, a); orc.step(a)
        env.sync()
        env.top_view.torch().zero_()                                        # what the call alone writes, all of it
        torch.cuda.synchronize()
        rcw.update_top_view_(env)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view, err_msg=f"stand-alone {kw}")
        env.close()
    env = rcw.SingleRoomModule.SingleRoom(batch=2, seed=1, **CFG1)           # no top view at all
    assert env.update_top_view_form() == "none"
    env.close()


def test_top_view_form_of_other_geometries(rcw):
    """What is not eligible for the two-kernel form keeps the one-kernel (LDS bit planes) or the in-place form.  Where a step's
    camera fill and the drawing go in one launch (256-row camera view, planes of a 256-thread draw workgroup) the two-kernel
    form is taken at EVERY batch size (round 4); where the drawing needs the side stream — here: a 128-row camera view —, only
    from 256 MiB of top view a step, unless rcw_set_top_view_form asks for it; a form the geometry cannot take is refused and
    the handle stays usable."""
    from raycastworlds_jl_amd import _capi

    for batch in (1, 512, 1024):                                            # 0.25 / 128 / 256 MiB: the fused launch at every size
        env = rcw.SingleRoomModule.SingleRoom(batch=batch, seed=1, render_top_view=True, pu_per_tu=32, **CFG2)
        assert env.top_view_form() == "two-kernels"
        env.set_top_view_form("one-kernel"); assert env.top_view_form() == "one-kernel"
        env.set_top_view_form(None); assert env.top_view_form() == "two-kernels"
        env.close()
    env = rcw.SingleRoomModule.SingleRoom(batch=512, seed=1, render_top_view=True, pu_per_tu=32, height_camera_view_pu=128, **CFG2)   # 128 MiB, side stream
    assert env.top_view_form() == "one-kernel"
    env.set_top_view_form("two-kernels"); assert env.top_view_form() == "two-kernels"
    rcw.act_(env, 1); env.sync()
    env.set_top_view_form(None); assert env.top_view_form() == "one-kernel"
    env.close()
    env = rcw.SingleRoomModule.SingleRoom(batch=1024, seed=1, render_top_view=True, pu_per_tu=32, height_camera_view_pu=128, **CFG2)  # 256 MiB
    assert env.top_view_form() == "two-kernels"
    env.close()
    for kw, form in ((dict(pu_per_tu=10, **CFG2), "two-kernels"),                     # 10 does not divide 256: the flat store kernel
                     (dict(pu_per_tu=12, height_tile_map_tu=16, width_tile_map_tu=8), "two-kernels"),   # 192 rows of 12-pixel tiles
                     (dict(pu_per_tu=32, player_radius_wu=0.49, position_increment_wu=0.1, **CFG2), "two-kernels"),   # circle of 33 rows
                     (dict(pu_per_tu=32, **CFG4), "two-kernels"),
                     (dict(pu_per_tu=13, height_tile_map_tu=9, width_tile_map_tu=9, num_rays=64), None),    # 117 rows: not a multiple of 4
                     (dict(pu_per_tu=6, height_tile_map_tu=10, width_tile_map_tu=9, num_rays=64), None),    # 6-pixel tiles, 60 rows
                     (dict(pu_per_tu=32, height_tile_map_tu=50, width_tile_map_tu=40, num_rays=128), None)):   # bit plane beyond LDS
        env = rcw.SingleRoomModule.SingleRoom(batch=2, seed=1, render_top_view=True, **kw)
        auto = env.top_view_form()
        if form is None:
            with pytest.raises(_capi.RcwError) as ei:
                env.set_top_view_form("two-kernels")
            assert ei.value.code == _capi.RCW_ERR_UNSUPPORTED
            assert env.top_view_form() == auto                               # the refused request changed nothing
        else:
            env.set_top_view_form("two-kernels")
            assert env.top_view_form() == form, kw
        rcw.act_(env, 1); env.sync()
        env.close()
    env = rcw.SingleRoomModule.SingleRoom(batch=2, seed=1, render_top_view=True, pu_per_tu=32, height_tile_map_tu=50,
                                          width_tile_map_tu=40, num_rays=128)
    assert env.top_view_form() == "in-place"
    with pytest.raises(_capi.RcwError):
        env.set_top_view_form("one-kernel")
    env.close()
    env = rcw.SingleRoomModule.SingleRoom(batch=2, seed=1, **CFG2)
    assert env.top_view_form() == "none"
    with pytest.raises(_capi.RcwError):
        env.set_top_view_form("one-kernel")
    env.close()


def test_two_top_view_handles_with_large_planes_alive_together(rcw, oracle):
    """Two render_top_view handles whose bit planes need more than 64 KiB of LDS (768² and 1024² px), created in the order
    that used to lower the first one's kernel limit, stepped alternately: the raised dynamic-LDS limit belongs to the
    kernel function, i.e. to every handle on the device, and is set once to the fixed cap."""
    big = dict(pu_per_tu=32, height_tile_map_tu=32, width_tile_map_tu=32, num_rays=128)
    small = dict(pu_per_tu=32, height_tile_map_tu=24, width_tile_map_tu=24, num_rays=128)
    e1, o1 = _make(rcw, oracle, 3, seed=5, render_top_view=1, out_of_bounds=1, **big)
    e2, o2 = _make(rcw, oracle, 3, seed=6, render_top_view=1, out_of_bounds=1, **small)
    rng = np.random.default_rng(4)
    for form in (None, "two-kernels"):
        e1.set_top_view_form(form); e2.set_top_view_form(form)
        for s in range(4):
            a = rng.integers(1, 5, 3).astype(np.uint8)
            rcw.act_(e1, a); o1.step(a)
            rcw.act_(e2, a); o2.step(a)
        np.testing.assert_array_equal(e1.top_view_host(), o1.top_view)
        np.testing.assert_array_equal(e2.top_view_host(), o2.top_view)
    e1.close(); e2.close()


@pytest.mark.parametrize("shape", ["fused", "side-stream", "one-kernel", "camera-only", "auto-reset"])
def test_captured_step_with_top_view_replays(rcw, oracle, shape):
    """A step is capturable into a HIP graph (torch.cuda.CUDAGraph on the stream the engine shares) in each of its launch
    shapes: the camera fill and the top view's drawing in one launch (the default), the drawing on the handle's side stream
    (a camera height the fused launch does not take: a fork and a join inside the capture), the one-kernel top view, the
    camera view alone, and with auto-reset on (episodes restart inside the replays, their generator keyed by the device-side
    episode counters): eight replays with the same device actions equal eight oracle steps, both images."""
    torch = pytest.importorskip("torch")
    kw = dict(render_top_view=1, pu_per_tu=32, out_of_bounds=1, **CFG2)
    if shape == "side-stream":
        kw["height_camera_view_pu"] = 128
    if shape in ("camera-only", "auto-reset"):
        kw = dict(out_of_bounds=1, **CFG2)
    if shape == "auto-reset":
        kw["auto_reset"] = True
    env, orc = _make(rcw, oracle, 48, seed=31, **kw)
    if shape in ("fused", "side-stream"):
        env.set_top_view_form("two-kernels")
        assert env.top_view_form() == "two-kernels"
    elif shape == "one-kernel":
        env.set_top_view_form("one-kernel")
    if shape == "auto-reset":                                                # every agent four forward steps from its goal: episodes end during the replays
        g = np.tile(np.array([[4, 6]], np.int32), (48, 1)); p = np.tile(np.array([[3.5, 4.5]], np.float32), (48, 1)); d = np.full(48, 32, np.int32)
        env.set_state(g, p, d); orc.set_state(g, p, d)
    stream = torch.cuda.Stream()
    env.set_stream(stream.cuda_stream)
    a_host = np.random.default_rng(2).integers(1, 5, env.batch).astype(np.uint8)
    with torch.cuda.stream(stream):
        actions = torch.from_numpy(a_host).cuda()
        for _ in range(2):                                                   # (warm-up outside the capture)
            rcw.act_(env, actions); orc.step(a_host)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            rcw.act_(env, actions)
        orc.step(a_host)                                                     # (torch runs nothing during capture; the engine's launches were captured, not run)
        g.replay(); stream.synchronize()
        for _ in range(7):
            g.replay(); orc.step(a_host)
        stream.synchronize()
    assert_state_equal(env, orc, where=f"after 8 graph replays ({shape})")
    if kw.get("render_top_view"):
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
    if shape == "auto-reset":
        np.testing.assert_array_equal(env.world.episode, orc.episode)
        assert (orc.episode > 0).any(), "no episode restarted inside the replays"
    env.close()


def test_drawing_first_and_camera_fill_on_the_side_stream(rcw, oracle, monkeypatch):
    """Round 5: where the camera fill is the shorter of the two, a step keeps the top view's DRAWING on the handle's stream (cast -> draw ->
    store, no event between them) and sends the camera fill to the side stream (rcw_api.hip, launch_top_view).  The rule takes it for big
    batches only (768^2 x 455 and 1024^2 x 256 in test_full_size_top_view run through it); here it is forced on small ones through the
    development build (RCW_TOP_DRAW_FIRST=1): plain steps, a masked reset, a change of stream, and a step captured into a HIP graph
    and replayed — both images and the state against the oracle every time."""
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("RCW_TOP_DRAW_FIRST", "1")
    rng = np.random.default_rng(21)
    for kw, batch in ((dict(pu_per_tu=32, height_camera_view_pu=128, **CFG2), 48), (dict(pu_per_tu=13, height_camera_view_pu=100, **CFG2), 37),
                      (dict(pu_per_tu=32, height_tile_map_tu=24, width_tile_map_tu=24, num_rays=128), 5)):
        env, orc = _make(rcw, oracle, batch, seed=17, render_top_view=1, out_of_bounds=1, library="dev", **kw)
        env.set_top_view_form("two-kernels")
        assert env.top_view_form() == "two-kernels"
        for s in range(4):
            a = rng.integers(1, 5, batch).astype(np.uint8)
            rcw.act_(env, a); orc.step(a)
        assert_state_equal(env, orc, where=f"drawing first, {kw}")
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        mask = (rng.random(batch) < 0.5).astype(np.uint8); mask[0] = 1
        rcw.reset_(env, mask=mask, seed=99); orc.reset(mask=mask, seed=99)
        a = rng.integers(1, 5, batch).astype(np.uint8)
        rcw.act_(env, a); orc.step(a)
        assert_state_equal(env, orc, where=f"drawing first, after a masked reset, {kw}")
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        stream = torch.cuda.Stream()
        env.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            actions = torch.from_numpy(a).cuda()
            rcw.act_(env, actions); orc.step(a)
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                rcw.act_(env, actions)
            for _ in range(5):
                g.replay(); orc.step(a)
            stream.synchronize()
        assert_state_equal(env, orc, where=f"drawing first, after 5 graph replays, {kw}")
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        env.close()


@pytest.mark.parametrize("parts", [2, 3, 4])
def test_several_draw_workgroups_an_agent(rcw, oracle, monkeypatch, parts):
    """Round 5: where a batch of big images leaves CUs without a draw workgroup, an agent's fan is split over 2 .. 4 workgroups that OR
    their planes into the agent's plane in HBM, and rcw_top_store_kernel leaves the plane zeroed behind it (rcw_top_draw_kernel,
    top_group_issue).  The rule takes it for few big images only (test_full_size_top_view: 1024^2 x 64 in four parts, 768^2 x 130 in
    two); here it is forced through the development build (RCW_TOP_PARTS) on small batches: steps, the stand-alone call, a masked
    reset (only the masked agents' planes are drawn and consumed), a change of form and back — every pixel against the oracle."""
    monkeypatch.setenv("RCW_TOP_PARTS", str(parts))
    rng = np.random.default_rng(30 + parts)
    for kw, batch in ((dict(pu_per_tu=32, height_camera_view_pu=128, **CFG2), 21), (dict(pu_per_tu=32, height_tile_map_tu=24, width_tile_map_tu=24, num_rays=128), 5),
                      (dict(pu_per_tu=32, height_camera_view_pu=300, num_rays=67, height_tile_map_tu=8, width_tile_map_tu=16), 9)):
        env, orc = _make(rcw, oracle, batch, seed=23, render_top_view=1, out_of_bounds=1, library="dev", **kw)
        env.set_top_view_form("two-kernels")
        for s in range(3):
            a = rng.integers(1, 5, batch).astype(np.uint8)
            rcw.act_(env, a); orc.step(a)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        env.sync(); env.top_view.torch().zero_()
        import torch
        torch.cuda.synchronize()
        rcw.update_top_view_(env)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        mask = (rng.random(batch) < 0.5).astype(np.uint8); mask[-1] = 1
        rcw.reset_(env, mask=mask, seed=5); orc.reset(mask=mask, seed=5)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        a = rng.integers(1, 5, batch).astype(np.uint8)
        rcw.act_(env, a); orc.step(a)
        assert_state_equal(env, orc, where=f"{parts} draw workgroups an agent, {kw}")
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        env.set_top_view_form("one-kernel"); rcw.act_(env, a); orc.step(a)
        env.set_top_view_form("two-kernels"); rcw.act_(env, a); orc.step(a)
        np.testing.assert_array_equal(env.top_view_host(), orc.top_view)
        env.close()


def test_ballot_bounded_march_gives_the_same_rays(rcw, oracle, monkeypatch):
    """The ballot-bounded march (the form north_star words; development switch RCW_CAST_MARCH=ballot of the development
    build librcw_hip_dev.so, measured against the shipped exec-masked march in profiles/) is the same function: bit-exact
    at the deep-march config and under every unpinned switch."""
    monkeypatch.setenv("RCW_CAST_MARCH", "ballot")
    rng = np.random.default_rng(12)
    env, orc = _make(rcw, oracle, 16, seed=9, library="dev", **CFG5)
    _rollout(rcw, env, orc, 30, rng, check_every=10, rays_every=10)
    env.close()
    for tie in (0, 1):
        for dist in (0, 1):
            env, orc = _make(rcw, oracle, 16, seed=5, dda_tie_break=tie, dda_distance=dist, library="dev", **CFG1)
            _rollout(rcw, env, orc, 30, rng, check_every=10, rays_every=10)
            env.close()
    env, orc = _make(rcw, oracle, 8, seed=3, T="Float64", library="dev", **CFG2)
    _rollout(rcw, env, orc, 20, rng, check_every=10, rays_every=10)
    env.close()


def test_lds_staged_ray_table_gives_the_same_rays(rcw, oracle, monkeypatch):
    """Development switch RCW_CAST_TABLE=lds of the development build (the heading's ray-table slice copied to LDS before
    use, as north_star words it; measured against the shipped direct L2 read in profiles/): same results."""
    monkeypatch.setenv("RCW_CAST_TABLE", "lds")
    rng = np.random.default_rng(14)
    for kw in (CFG2, CFG5, dict(T="Float64", **CFG1), dict(num_rays=100, height_tile_map_tu=9, width_tile_map_tu=7)):
        env, orc = _make(rcw, oracle, 12, seed=4, auto_reset=True, library="dev", **kw)
        _rollout(rcw, env, orc, 25, rng, check_every=5, rays_every=5)
        env.close()


@pytest.mark.parametrize("form", ["1", "2"])
def test_whole_step_in_one_launch_gives_the_same_frames(rcw, oracle, monkeypatch, form):
    """Development switch RCW_STEP_FUSED of the development build (measured and rejected, docs/experiments.md): cast and camera fill
    in ONE launch, the column descriptors handed from the casting wavefronts to the fill workgroups inside it — through a
    per-agent flag (1) or through words that carry the step's epoch (2).  Same frames, same state, with masked resets and
    auto-reset; also in Float64 and with a batch that is not a multiple of a workgroup's four agents."""
    monkeypatch.setenv("RCW_STEP_FUSED", form)
    rng = np.random.default_rng(21)
    for batch, kw in ((203, dict(num_rays=100, height_tile_map_tu=9, width_tile_map_tu=7)), (64, dict(T="Float64", **CFG1)),
                      (1030, dict(num_rays=256, height_tile_map_tu=8, width_tile_map_tu=8))):
        env, orc = _make(rcw, oracle, batch, seed=6, auto_reset=True, out_of_bounds=1, library="dev", **kw)
        assert env.fill_kernel_name() == "rcw_step256_kernel"
        for s in range(12):
            if s % 5 == 2:
                mask = (rng.random(batch) < 0.4).astype(np.uint8)
                rcw.reset_(env, mask=mask, seed=40 + s); orc.reset(mask=mask, seed=40 + s)
            a = rng.integers(1, 5, batch).astype(np.uint8)
            rcw.act_(env, a)
            assert orc.step(a) == 0
            if s % 4 == 3 or s == 11:
                assert_state_equal(env, orc, frames=True, where=f"fused form {form}, {kw}, step {s}")
        env.close()


def test_create_destroy_cycles_leave_device_memory_unchanged(rcw):
    """Handles with every optional buffer (top view, Float64 tables, typed rewards, rays scratch, a bound observation
    buffer, the gather scratch is covered by the RCCL test) are created, used and destroyed 25 times: free device
    memory returns to where it was (hipMalloc'd memory is released, not cached, by rcw_destroy)."""
    torch = pytest.importorskip("torch")
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)

    def cycle(k):
        env = rcw.SingleRoomModule.SingleRoom(batch=192, seed=k, render_top_view=True, R=("Float64", "Int64")[k & 1],
                                              T=("Float32", "Float64")[(k >> 1) & 1], auto_reset=True, **CFG2)
        for _ in range(3):
            rcw.act_(env, rng.integers(1, 5, 192).astype(np.uint8))
        env.world.rays(0, 64)
        env.top_view_host(0, 2)
        rcw.update_camera_view_(env); rcw.update_top_view_(env)
        env.sync()
        env.close()

    for k in range(4):                              # first use of each type combination loads its kernels' code objects etc.
        cycle(k)
    # an environment runs on a torch.cuda.Stream() when torch is loaded: torch hands those out of a pool of 32 per device that it
    # fills on demand (a new HIP stream, ~1 MiB of device memory, for each of the first 32 requests — tools/_build/memdrift.py
    # measured +1 MiB per environment up to the 32nd and nothing after it): fill the pool before measuring
    # (a HIP stream gets its queue with its first command: run something on each)
    for _ in range(32):
        with torch.cuda.stream(torch.cuda.Stream()):
            torch.zeros(8, device="cuda").add_(1)
    gc.collect()                                    # (what earlier tests dropped without close() goes now, not during the cycles)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(25):
        cycle(k)
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # a leak makes free memory SHRINK; it may grow (the suite's earlier garbage finalised late: seen once, +48 MiB)
    assert free0 - free1 <= 8 << 20, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 25 cycles"


def test_dropped_environment_is_finalised_and_its_memory_returned(rcw):
    """with the RLBase tensors made (reward / is_terminated / state, the cached aliases), an environment
    that is simply dropped — no close() — is collected and its device memory returned; while one of its tensors lives,
    the engine's memory stays valid (the tensor's exporter holds the native handle, nothing holds the environment)."""
    import gc
    import weakref

    torch = pytest.importorskip("torch")
    RL = rcw.RLBase
    torch.cuda.synchronize()

    def make():
        env = rcw.SingleRoomModule.SingleRoom(batch=1024, seed=1, **CFG2)          # 256 MiB of frames
        rl = rcw.RLBaseEnv(env)
        r = RL.reward(rl).torch(sync=False)
        d = RL.is_terminated(rl).torch(sync=False)
        s = RL.state(rl).torch(sync=False)
        rl(3)
        return env, r, d, s

    env, r, d, s = make()
    env.close(); del env, r, d, s                       # code objects loaded, allocator warm
    gc.collect(); torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    env, r, d, s = make()
    ref = weakref.ref(env)
    assert torch.cuda.mem_get_info()[0] < free0 - (200 << 20)
    del env, d, s
    gc.collect()
    assert ref() is None, "the environment is kept alive by its cached tensors"
    assert torch.cuda.mem_get_info()[0] < free0 - (200 << 20), "the engine was destroyed under a live tensor"
    assert float(r.sum()) == 0.0                        # still readable: turning left earns nothing
    del r
    gc.collect(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free1 - free0) <= 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB not returned"


def test_external_stream_is_not_record_streamed(rcw, oracle):
    """a torch.cuda.ExternalStream wraps a stream its CREATOR owns; the engine must hold tensors by
    reference there (the event ring), never Tensor.record_stream — torch would touch the stream again whenever the
    tensor is freed, possibly after its owner destroyed it (profiles/r03_record_stream_abort.txt)."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 64, seed=23, **CFG1)
    own = torch.cuda.Stream()
    env.set_stream(own)
    assert env._stream_is_torch_owned()
    ext = torch.cuda.ExternalStream(own.cuda_stream)    # the same stream, seen as somebody else's
    env.set_stream(ext)
    assert env.stream_ptr() == own.cuda_stream and not env._stream_is_torch_owned()
    rng = np.random.default_rng(2)
    for _ in range(30):
        a = rng.integers(1, 5, 64).astype(np.uint8)
        rcw.act_(env, torch.from_numpy(a).cuda())       # a temporary on torch's current stream: cross-stream hand-over
        orc.step(a)
    assert 0 < len(env._held) <= 16, "the temporaries were not held by reference"
    assert_state_equal(env, orc, where="on an ExternalStream")
    env.close()


def test_rng_keyword_resets_from_the_callers_generator(rcw, oracle):
    """SingleRoom(; rng) / reset!(env) with the caller's generator (SR:49,265; draws at SR:62-74, SR:120-128, UT:24-28): the
    engine's state after construction, after reset_(env), after a masked reset and after a reset from per-agent generators
    equals the oracle given the draws a TWIN generator (same seed) yields in the reference's order."""
    SR = rcw.SingleRoomModule
    B, H, W, nd = 48, 8, 16, 128

    def expect(orc, gens, mask=None, construction=False):
        goal, pos, head = orc.goal.copy(), orc.position.copy(), orc.direction.copy()
        for a in range(B):
            if mask is not None and not mask[a]:
                continue
            g = gens[a] if isinstance(gens, list) else gens
            if construction:
                SR.reference_reset_draws(g, H, W, nd)
            gi, gj, ti, tj, d = SR.reference_reset_draws(g, H, W, nd)
            goal[a], pos[a], head[a] = (gi, gj), (ti - 0.5, tj - 0.5), d
        orc.set_state(goal, pos, head, mask=mask)

    env = SR.SingleRoom(batch=B, rng=np.random.default_rng(42), out_of_bounds=1, num_rays=64)     # the reference's default 8 x 16 room
    twin = np.random.default_rng(42)
    orc = oracle.OracleBatch(B, out_of_bounds=1, num_rays=64)
    expect(orc, twin, construction=True)
    assert_state_equal(env, orc, rays=True, where="after construction with rng")
    rng = np.random.default_rng(3)
    _rollout(rcw, env, orc, 30, rng, check_every=10)
    rcw.reset_(env)                                      # the environment's own rng goes on where it stopped
    expect(orc, twin)
    assert_state_equal(env, orc, where="after reset!(env) with the environment's rng")
    assert (env.world.reward == 0).all() and not env.world.done.any()
    mask = (rng.random(B) < 0.4).astype(np.uint8); mask[5] = 1
    rcw.reset_(env, mask=mask)
    expect(orc, twin, mask=mask)
    assert_state_equal(env, orc, where="after a masked reset with rng")
    _rollout(rcw, env, orc, 10, rng, check_every=5)
    rcw.reset_(env, rng=[np.random.default_rng(100 + a) for a in range(B)])       # one generator per agent
    expect(orc, [np.random.default_rng(100 + a) for a in range(B)])
    assert_state_equal(env, orc, where="after a reset from per-agent generators")
    # agent a is then the reference world built with rng[a]: agents with equal generators are equal
    rcw.reset_(env, rng=[np.random.default_rng(7) for _ in range(B)])
    assert len({tuple(r) for r in env.world.goal_position.tolist()}) == 1
    env.close()


def test_play_keys_replays_the_keyboard_callback(rcw, oracle, tmp_path):
    """Headless `play!` (SR:488-568): a scripted key sequence — W/S/A/D act, R resets, V toggles the view, an unbound
    key warns, Q closes — with the frame buffer after every key compared with the oracle's views blitted the
    reference's way (camera view 64 x 256 and top view 256 x 256 in a 256 x 256 MiniFB buffer)."""
    env, orc = _make(rcw, oracle, 3, seed=21, render_top_view=1, out_of_bounds=1, **CFG1)
    keys = "wwadVwsVxdRwwQww"
    with pytest.warns(UserWarning, match="No keybinding exists for x"):
        out = rcw.play_keys(env, keys, agent=1, frame_dir=str(tmp_path))
    assert [o[0] for o in out] == list("wwadvwsvxdrww")          # Q closes: nothing after it runs
    view, steps = 1, 0
    for key, steps_taken, reward, done, fb in out:
        if key in "wsad":
            orc.step(np.full(3, "wsad".index(key) + 1, dtype=np.uint8)); steps += 1
        elif key == "r":
            orc.reset(seed=21); steps = 0
        elif key == "v":
            view = 2 if view == 1 else 1
        image = orc.camera_view[1] if view == 1 else orc.top_view[1]
        want = np.zeros((256, 256), dtype=np.uint32)
        want[: image.shape[1], : image.shape[0]] = image.T
        np.testing.assert_array_equal(fb, want, err_msg=f"frame buffer after key {key!r}")
        assert steps_taken == steps and reward == orc.reward[1] and done == bool(orc.done[1])
    frames = sorted(os.listdir(tmp_path))
    assert len(frames) == len(out) and frames[0] == "frame_0000.ppm"
    assert open(tmp_path / frames[0], "rb").read(15) == b"P6\n256 256\n255\n"
    env.close()
