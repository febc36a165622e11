"""GPU parity of the step's two forms (include/rcw.h: rcw_step_form / rcw_set_step_form).

RCW.act!(env, a) SR:333-340 as ONE launch — the fill workgroups pick each agent's frame among the successor states the previous
launch cast (rcw_fill256_cast_kernel) — and as the cast kernel followed by the fill kernel must leave the same state and the same
pixels as the CPU oracle, step by step: through masked resets and injected states (which cast as a launch of their own and leave
the touched agents' slots ready), episodes that restart under auto_reset (the re-sampled world is drawn ahead without being
committed), moves that are blocked / reach the goal / raise (SR:162-176: the successor is the current frame), invalid device
actions (the agent keeps its frame), both world-unit types, both casting shapes (a wavefront per agent up to 256 view columns, a
workgroup per agent beyond; more than 1024 columns: the table tail), a change of form in the middle of a rollout, a rebound
observation buffer, and a captured step (which turns the handle to the two-launch form).
"""
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
        okw["world_unit_bits"] = 64
    orc = oracle.OracleBatch(batch, seed=seed, **okw)
    return env, orc


def _steps(rcw, env, orc, n, rng, every=1, where=""):
    for s in range(n):
        a = rng.integers(1, 5, env.batch).astype(np.uint8)
        rcw.act_(env, a)
        assert orc.step(a) == 0
        if (s + 1) % every == 0 or s == n - 1:
            assert_state_equal(env, orc, where=f"{where} step {s}")


def test_the_rule_and_the_refusals(rcw):
    from raycastworlds_jl_amd import _capi

    SR = rcw.SingleRoomModule.SingleRoom
    with SR(batch=1024, seed=1, **CFG2) as env:                              # 256 MiB of frames a step: the fill outlasts the casting half's five fans
        assert env.step_form() == "one-launch" and env.fill_kernel_name() == "rcw_fill256_cast_kernel"
        env.set_step_form("two-launches")
        assert env.step_form() == "two-launches" and env.fill_kernel_name() == "rcw_fill256_kernel"
        env.set_step_form(None)
        assert env.step_form() == "one-launch"
        with pytest.raises(ValueError):
            env.set_step_form("three")
    # ... and the batch must be large enough for it to pay (profiles/r06_small_batches.txt): by the rule a small batch keeps the two launches,
    # on request it takes the one
    for kw, small, large in ((CFG2, 256, 768), (CFG1, 512, 2048), (CFG3, 128, 512), (CFG5, 256, 512)):
        with SR(batch=small, seed=1, **kw) as env:
            assert env.step_form() == "two-launches", (kw, small)
            env.set_step_form("one-launch"); assert env.step_form() == "one-launch"
            env.set_step_form(None); assert env.step_form() == "two-launches"
        with SR(batch=large, seed=1, **kw) as env:
            assert env.step_form() == "one-launch", (kw, large)
    # the other camera heights of the moving window (rcw_fill_window_kernel's: 256 k, 128, 64 rows) take it too
    for hc in (64, 128, 512, 768, 1024):
        with SR(batch=4, seed=1, height_camera_view_pu=hc, **CFG1) as env:
            assert env.step_form() == "two-launches"                        # (4 agents: the rule)
            env.set_step_form("one-launch"); assert env.step_form() == "one-launch" and env.fill_kernel_name() == "rcw_fill_window_cast_kernel"
    # what the one-launch form does not take: a camera height of the flat / frame kernels, a slot word's 13 bits of padding, the top view (its
    # drawing needs the state the same launch commits)
    for kw in (dict(height_camera_view_pu=100, **CFG1), dict(height_camera_view_pu=20, **CFG1), dict(height_camera_view_pu=8192, **CFG1),
               dict(height_camera_view_pu=128, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=33), dict(render_top_view=1, **CFG1)):
        with SR(batch=4, seed=1, **kw) as env:
            assert env.step_form() == "two-launches"
            with pytest.raises(_capi.RcwError) as e:
                env.set_step_form("one-launch")
            assert e.value.code == -7                                      # RCW_ERR_UNSUPPORTED
            assert env.step_form() == "two-launches"
            env.set_step_form("two-launches"); env.set_step_form(None)


SHAPES = [
    ("cfg1", dict(**CFG1), 5),
    ("cfg2, a wavefront per agent", dict(out_of_bounds=1, **CFG2), 67),                   # (67: the last casting workgroup is short)
    ("reference default", dict(**REFERENCE_DEFAULT), 9),
    ("100 columns", dict(height_tile_map_tu=7, width_tile_map_tu=11, num_rays=100, num_directions=96), 13),
    ("cfg3, a workgroup per agent", dict(**CFG3), 6),
    ("cfg4", dict(**CFG4), 10),
    ("cfg5", dict(**CFG5), 3),
    ("1500 columns: the table's tail", dict(height_tile_map_tu=9, width_tile_map_tu=8, num_rays=1500, num_directions=64), 3),
    ("Float64", dict(T="Float64", **CFG2), 7),
    ("Float64, 600 columns", dict(T="Float64", height_tile_map_tu=8, width_tile_map_tu=8, num_rays=600), 3),
    ("the other march", dict(dda_tie_break=1, dda_distance=1, normalize_mode=1, **CFG1), 6),
    ("inc > radius", dict(position_increment_wu=0.25, player_radius_wu=0.125, out_of_bounds=0, **CFG1), 12),
    ("128 rows (two columns a chunk)", dict(height_camera_view_pu=128, **CFG2), 9),
    ("64 rows (four columns a chunk), 100 columns", dict(height_camera_view_pu=64, height_tile_map_tu=7, width_tile_map_tu=11, num_rays=100), 11),
    ("512 rows, a workgroup per agent", dict(height_camera_view_pu=512, height_tile_map_tu=9, width_tile_map_tu=9, num_rays=300), 5),
    ("768 rows (three chunks a column), Float64", dict(T="Float64", height_camera_view_pu=768, **CFG1), 6),
    ("1024 rows", dict(height_camera_view_pu=1024, height_tile_map_tu=6, width_tile_map_tu=6, num_rays=33), 7),
]


@pytest.mark.parametrize("auto_reset", [False, True])
@pytest.mark.parametrize("name,kw,batch", SHAPES, ids=[s[0] for s in SHAPES])
def test_both_forms_match_the_oracle_step_by_step(rcw, oracle, name, kw, batch, auto_reset):
    """Every step compared; in the middle: a masked reset, an injected state, a change of form, a rebound observation buffer."""
    for form in ("one-launch", "two-launches"):
        env, orc = _make(rcw, oracle, batch, seed=21, auto_reset=auto_reset, **kw)
        env.set_step_form(form)
        assert env.step_form() == form
        rng = np.random.default_rng(5)
        lenient = kw.get("out_of_bounds", 0) == 0

        def steps(n, where):
            for s in range(n):
                a = rng.integers(1, 5, batch).astype(np.uint8)
                rcw.act_(env, a)
                assert orc.step(a) == 0
                try:
                    env.sync()
                except IndexError:                                          # the reference raised for some agent (RCW_OOB_ERROR): same status words, then on
                    assert lenient
                    np.testing.assert_array_equal(env.world.status, orc.status)
                    env.clear_error(); orc.clear_status()
                assert_state_equal(env, orc, where=f"{name} / {form} / {where} step {s}")

        assert_state_equal(env, orc, where=f"{name} / {form} after create")
        steps(14, "first")
        mask = (rng.random(batch) < 0.5).astype(np.uint8); mask[0] = 1; mask[-1] = 0
        rcw.reset_(env, mask=mask, seed=77); orc.reset(mask=mask, seed=77)
        assert_state_equal(env, orc, where=f"{name} / {form} after a masked reset")
        steps(8, "after the masked reset")
        # an injected state for every other agent, right in front of a wall, looking at it: blocked moves from the first step on
        H, W = orc.H, orc.W
        goal = np.tile(np.array([[2, 2]], np.int32), (batch, 1))
        pos = np.tile(np.array([[H - 1 - 0.25, 1.5 + (W - 3) / 2]], np.float64 if kw.get("T") == "Float64" else np.float32), (batch, 1))
        d = np.zeros(batch, np.int32)
        m2 = np.zeros(batch, np.uint8); m2[::2] = 1
        env.set_state(goal, pos, d, mask=m2); orc.set_state(goal, pos, d, mask=m2)
        assert_state_equal(env, orc, where=f"{name} / {form} after set_state")
        steps(6, "after set_state")
        env.set_step_form("two-launches" if form == "one-launch" else "one-launch")
        steps(6, "after the change of form")
        env.set_step_form(form)
        steps(5, "back")
        if auto_reset:
            np.testing.assert_array_equal(env.world.episode, orc.episode)
        env.close()


def test_episodes_restart_in_the_one_launch_form(rcw, oracle):
    """Agents four forward moves from their goal, auto_reset on: the goal is reached (done, reward 1, the frame stays), the next action
    — whatever it is — re-samples the agent, and the frame of the re-sampled world was cast one launch ahead (reset_preview) without
    the episode counter, the goal or the tile map having moved: every step's frames, tile maps and episode counters against the oracle."""
    B = 96
    for kw in (dict(out_of_bounds=1, **CFG2), dict(out_of_bounds=1, **CFG3), dict(T="Float64", out_of_bounds=1, **CFG1)):
        env, orc = _make(rcw, oracle, B, seed=3, auto_reset=True, **kw)
        env.set_step_form("one-launch")
        assert env.step_form() == "one-launch"
        f = np.float64 if kw.get("T") == "Float64" else np.float32
        g = np.tile(np.array([[4, 6]], np.int32), (B, 1)); p = np.tile(np.array([[3.5, 4.5]], f), (B, 1)); d = np.full(B, 32, np.int32)
        env.set_state(g, p, d); orc.set_state(g, p, d)
        rng = np.random.default_rng(8)
        restarted = 0
        for s in range(40):
            a = np.where(rng.random(B) < 0.7, 1, rng.integers(1, 5, B)).astype(np.uint8)
            rcw.act_(env, a); assert orc.step(a) == 0
            assert_state_equal(env, orc, where=f"restart rollout {kw} step {s}")
            np.testing.assert_array_equal(env.world.episode, orc.episode)
            restarted = int(orc.episode.sum())
        assert restarted > B // 2, "too few episodes restarted for the test to mean anything"
        env.close()


def test_invalid_device_actions_keep_the_frame_in_the_one_launch_form(rcw, oracle):
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 40, seed=4, **CFG2)
    env.set_step_form("one-launch")
    assert env.step_form() == "one-launch"
    rng = np.random.default_rng(9)
    for s in range(12):
        a = rng.integers(1, 5, env.batch).astype(np.uint8)
        if s % 3 == 1:
            a[[3, 17, 39]] = [0, 9, 255]
        rcw.act_(env, torch.from_numpy(a).cuda())
        if s % 3 == 1:
            orc.step_lenient(a)
            with pytest.raises(AssertionError):
                env.sync()
            np.testing.assert_array_equal(env.world.status, orc.status)
            env.clear_error(); orc.clear_status()
        else:
            assert orc.step(a) == 0
        assert_state_equal(env, orc, where=f"invalid device actions, step {s}")
    env.close()


def test_rebound_observation_buffer_and_separate_renderers(rcw, oracle):
    """rcw_bind_obs alternates two caller buffers between one-launch steps; cast_rays! / update_camera_view! alone leave the slots alone."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 24, seed=6, **CFG2)
    env.set_step_form("one-launch")
    bufs = [torch.zeros((24, 256, 256), dtype=torch.int32, device="cuda") for _ in range(2)]
    rng = np.random.default_rng(1)
    for s in range(10):
        env.bind_obs(bufs[s & 1].data_ptr())
        a = rng.integers(1, 5, 24).astype(np.uint8)
        rcw.act_(env, a); assert orc.step(a) == 0
        env.sync()
        np.testing.assert_array_equal(bufs[s & 1].cpu().numpy().view(np.uint32), orc.camera_view, err_msg=f"bound buffer, step {s}")
        if s == 4:
            rcw.cast_rays_(env); rcw.update_camera_view_(env)
    env.bind_obs(None)
    _steps(rcw, env, orc, 4, rng, where="library buffer again")
    env.close()


def test_a_captured_step_turns_the_handle_to_two_launches(rcw, oracle):
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 48, seed=31, out_of_bounds=1, auto_reset=True, **CFG2)
    env.set_step_form("one-launch")
    assert env.step_form() == "one-launch"
    stream = torch.cuda.Stream()
    env.set_stream(stream.cuda_stream)
    a_host = np.random.default_rng(2).integers(1, 5, env.batch).astype(np.uint8)
    with torch.cuda.stream(stream):
        actions = torch.from_numpy(a_host).cuda()
        for _ in range(3):
            rcw.act_(env, actions); orc.step(a_host)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            rcw.act_(env, actions)
        assert env.step_form() == "two-launches"
        for _ in range(6):
            g.replay(); orc.step(a_host)
        stream.synchronize()
        assert_state_equal(env, orc, where="after 6 replays")
        # direct steps after the replays: still two launches under the rule (the slots went stale behind the library's back) ...
        for _ in range(3):
            rcw.act_(env, actions); orc.step(a_host)
        assert env.step_form() == "two-launches"
        assert_state_equal(env, orc, where="direct steps after the replays")
        env.set_step_form(None)
        assert env.step_form() == "two-launches"
        # ... until the caller asks for the one-launch form again (no replay from here on: the caller's promise)
        env.set_step_form("one-launch")
        for _ in range(5):
            rcw.act_(env, actions); orc.step(a_host)
        stream.synchronize()
    assert env.step_form() == "one-launch"
    assert_state_equal(env, orc, where="one launch again")
    env.close()


def test_descriptors_are_refreshed_on_demand_and_live_once_their_pointers_are_out(rcw, oracle):
    """The one-launch step leaves the (height_line_pu, colour id) descriptors alone unless somebody holds their device pointers
    (include/rcw.h, rcw_columns): rcw_columns recasts on demand; after rcw_columns_device_ptr every step refreshes them in place."""
    torch = pytest.importorskip("torch")
    env, orc = _make(rcw, oracle, 33, seed=12, out_of_bounds=1, **CFG2)
    env.set_step_form("one-launch")
    assert env.step_form() == "one-launch"
    rng = np.random.default_rng(4)
    for s in range(6):
        a = rng.integers(1, 5, 33).astype(np.uint8)
        rcw.act_(env, a); assert orc.step(a) == 0
    h, c = env.columns()                                                    # on demand, after six steps that did not write them
    np.testing.assert_array_equal(h, orc.col_height); np.testing.assert_array_equal(c, orc.col_colour)
    a = rng.integers(1, 5, 33).astype(np.uint8)
    rcw.act_(env, a); assert orc.step(a) == 0
    rcw.update_camera_view_(env)                                            # update_camera_view! alone fills from them: recast first
    np.testing.assert_array_equal(env.camera_view_host(), orc.camera_view)
    hd, cd = env.columns_device()                                           # from here on: live
    for s in range(5):
        a = rng.integers(1, 5, 33).astype(np.uint8)
        rcw.act_(env, a); assert orc.step(a) == 0
        env.sync()
        np.testing.assert_array_equal(hd.torch().cpu().numpy(), orc.col_height, err_msg=f"live descriptors, step {s}")
        np.testing.assert_array_equal(cd.torch().cpu().numpy(), orc.col_colour, err_msg=f"live descriptors, step {s}")
    mask = np.zeros(33, np.uint8); mask[::3] = 1
    rcw.reset_(env, mask=mask, seed=5); orc.reset(mask=mask, seed=5)
    assert_state_equal(env, orc, where="masked reset with live descriptors")
    env.close()
    # ... and a masked reset BEHIND steps that left them stale: the masked agents' are fresh, the others' recast on demand
    env, orc = _make(rcw, oracle, 33, seed=13, out_of_bounds=1, **CFG2)
    env.set_step_form("one-launch")
    for s in range(4):
        a = rng.integers(1, 5, 33).astype(np.uint8)
        rcw.act_(env, a); assert orc.step(a) == 0
    rcw.reset_(env, mask=mask, seed=6); orc.reset(mask=mask, seed=6)
    assert_state_equal(env, orc, where="masked reset behind stale descriptors")
    env.close()


def test_a_masked_reset_with_a_new_seed_reaches_the_agents_it_does_not_touch(rcw, oracle):
    """The reset generator's seed is the handle's (rcw_reset sets it for every agent): an agent that is DONE under auto_reset and is NOT in a
    masked reset's mask is re-sampled by its next action with the NEW seed.  The one-launch step had cast that agent's successors from a
    preview drawn with the old seed: rcw_reset must not leave those slots in place.  Every agent one forward move from its goal, one step
    (all done), a masked reset of half of them with another seed, then steps: frames, poses, tile maps and episode counters against the oracle."""
    B = 48
    for kw in (dict(out_of_bounds=1, **CFG2), dict(out_of_bounds=1, **CFG3)):
        env, orc = _make(rcw, oracle, B, seed=5, auto_reset=True, **kw)
        env.set_step_form("one-launch")
        g = np.tile(np.array([[4, 6]], np.int32), (B, 1)); p = np.tile(np.array([[3.5, 4.875]], np.float32), (B, 1)); d = np.full(B, 32, np.int32)
        env.set_state(g, p, d); orc.set_state(g, p, d)
        fwd = np.ones(B, np.uint8)
        rcw.act_(env, fwd); assert orc.step(fwd) == 0
        assert orc.done.all(), "the set-up did not bring the agents to their goal"
        assert_state_equal(env, orc, where=f"all done {kw}")
        mask = np.zeros(B, np.uint8); mask[::2] = 1
        rcw.reset_(env, mask=mask, seed=4242); orc.reset(mask=mask, seed=4242)
        assert_state_equal(env, orc, where=f"masked reset with a new seed {kw}")
        rng = np.random.default_rng(3)
        for s in range(6):
            a = rng.integers(1, 5, B).astype(np.uint8)
            rcw.act_(env, a); assert orc.step(a) == 0
            assert_state_equal(env, orc, where=f"step {s} behind the masked reset {kw}")
            np.testing.assert_array_equal(env.world.episode, orc.episode)
        assert env.step_form() == "one-launch"
        env.close()
