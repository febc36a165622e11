"""The reference's own test (test/runtests.jl:15-44), restated per agent: 5 resets x up to 5000
uniformly random actions at the default config (8 x 16 map, 512 rays, R = Float32); after a
reset reward == 0 and not terminated; on the step where an agent terminates its episode
return equals goal_reward; no step raises.  CPU: against the oracle.  GPU: through the C ABI
with the RLBase verbs, read like the Julia test."""
import numpy as np
import pytest

MAX_STEPS = 5000      # runtests.jl:6
NUM_RESETS = 5        # runtests.jl:7


def test_oracle_random_policy_invariants(oracle):
    B = 16
    orc = oracle.OracleBatch(B, seed=3, render=False, out_of_bounds=1)   # default config
    rng = np.random.default_rng(0)
    terminated = 0
    for r in range(NUM_RESETS):
        orc.reset(seed=100 + r)
        assert (orc.reward == 0).all() and not orc.done.any()            # runtests.jl:22-23
        total = np.zeros(B, np.float32)
        live = np.ones(B, bool)
        for i in range(600):
            a = rng.integers(1, 5, B)
            assert orc.step(a) == 0
            total += np.where(live, orc.reward, 0)
            fin = live & (orc.done != 0)
            assert (total[fin] == 1.0).all()                             # runtests.jl:33
            assert (orc.reward[~(orc.done != 0)] == 0).all()             # non-terminal steps pay nothing
            terminated += int(fin.sum())
            live &= ~fin
            if not live.any():
                break
        assert (orc.status == 0).all()
    assert terminated > 0


def test_oracle_auto_reset_and_lenient_step_invariants(oracle):
    """The checker's own two batch conventions, by what must hold whatever the generator draws (CPU; `make -C oracle cov` showed
    no CPU test running these lines — the GPU parity tests compare the HIP path WITH them): with auto_reset an agent that was
    terminated is reset by the next step INSTEAD of acting — a new episode, reward 0, not terminated, the player at the centre
    of an empty interior tile, the goal inside the wall ring (SR:110-137) — and the others act as without it; the lenient step
    (rcw_step_device's convention) skips an agent with an invalid action, flags it, and steps the rest."""
    B, H, W = 48, 8, 8
    kw = dict(height_tile_map_tu=H, width_tile_map_tu=W, num_rays=16, render=False, out_of_bounds=1)
    auto, plain = oracle.OracleBatch(B, seed=9, auto_reset=1, **kw), oracle.OracleBatch(B, seed=9, auto_reset=0, **kw)
    rng = np.random.default_rng(2)
    restarts = 0
    for s in range(1500):
        was_done = auto.done.copy() != 0
        before = (auto.position.copy(), auto.direction.copy(), auto.goal.copy(), auto.episode.copy())
        # keep the twin without auto-reset in the same state, so that "the others act as without it" can be checked
        plain.set_state(before[2], before[0], before[1])
        a = rng.integers(1, 5, B).astype(np.uint8)
        assert auto.step(a) == 0 and plain.step(a) == 0
        ep = auto.episode
        assert (ep[was_done] == before[3][was_done] + 1).all() and (ep[~was_done] == before[3][~was_done]).all()
        assert (auto.reward[was_done] == 0).all() and not auto.done[was_done].any()          # SR:131-132
        pos, goal = auto.position[was_done], auto.goal[was_done]
        tile = np.floor(pos).astype(int) + 1                                                  # wu_to_tu  UT:5
        assert (pos == tile - 0.5).all()                                                      # the centre of a tile  SR:125
        assert ((tile >= 2) & (tile <= [H - 1, W - 1])).all() and ((goal >= 2) & (goal <= [H - 1, W - 1])).all()
        assert not (tile == goal).all(axis=1).any()                                           # an EMPTY tile  UT:27
        assert ((auto.direction[was_done] >= 0) & (auto.direction[was_done] < 128)).all()     # SR:128
        keep = ~was_done
        np.testing.assert_array_equal(auto.position[keep], plain.position[keep])
        np.testing.assert_array_equal(auto.direction[keep], plain.direction[keep])
        np.testing.assert_array_equal(auto.done[keep], plain.done[keep])
        np.testing.assert_array_equal(auto.col_height[keep], plain.col_height[keep])
        restarts += int(was_done.sum())
    assert restarts >= 8, restarts
    # the lenient step
    len_, ref = oracle.OracleBatch(B, seed=11, **kw), oracle.OracleBatch(B, seed=11, **kw)
    for s in range(40):
        a = rng.integers(1, 5, B).astype(np.uint8)
        bad = rng.random(B) < 0.2
        a_bad = np.where(bad, rng.choice([0, 5, 200], B), a).astype(np.uint8)
        before = (len_.position.copy(), len_.direction.copy(), len_.goal.copy())
        ref.set_state(before[2], before[0], before[1])
        len_.clear_status()
        assert len_.step_lenient(a_bad) == 0 and ref.step(a) == 0
        assert (len_.status[bad] == -2).all() and (len_.status[~bad] == 0).all()              # RCW_ERR_INVALID_ACTION on the skipped ones only
        np.testing.assert_array_equal(len_.position[bad], before[0][bad])
        np.testing.assert_array_equal(len_.direction[bad], before[1][bad])
        np.testing.assert_array_equal(len_.position[~bad], ref.position[~bad])
        np.testing.assert_array_equal(len_.direction[~bad], ref.direction[~bad])
        np.testing.assert_array_equal(len_.col_height[~bad], ref.col_height[~bad])
        assert ref.step(a_bad) == -2                                                          # the strict step rejects the batch (SR:140's @assert)
        ref.clear_status()


@pytest.mark.gpu
def test_rlbase_random_policy_invariants(rcw):
    RLBase = rcw.RLBase
    B = 256
    env = rcw.RLBaseEnv(rcw.SingleRoomModule.SingleRoom(batch=B, seed=3, out_of_bounds=1))   # Env(R = R)
    rng = np.random.default_rng(0)
    terminated = 0
    for r in range(NUM_RESETS):
        RLBase.reset_(env)
        assert (RLBase.reward(env) == 0).all()                           # runtests.jl:22
        assert not RLBase.is_terminated(env).any()                       # runtests.jl:23
        total = np.zeros(B, np.float32)
        live = np.ones(B, bool)
        for i in range(MAX_STEPS):
            state = RLBase.state(env)                                    # runtests.jl:27
            assert state.shape == (B, 512, 256)
            action = rng.integers(1, 5, B)                               # rand(action_space(env))
            env(action)
            rew, term = RLBase.reward(env), RLBase.is_terminated(env)
            total += np.where(live, rew, 0)
            fin = live & term
            assert (total[fin] == 1.0).all()                             # total_reward in terminal returns
            terminated += int(fin.sum())
            live &= ~fin
            if not live.any() or i >= 800:                               # bounded for the GPU budget
                break
    assert terminated > 0, "no agent out of 256 reached the goal in 5 x 800 random steps"
    env.env.close()


def test_sampler_gives_up_on_a_map_without_an_empty_tile(oracle):
    """utils.jl:23-37: after max_tries = 1024 H W occupied draws `sample_empty_position` @warns (utils.jl:34) and returns the occupied
    tile; reset! goes on with it.  Only a map with no empty tile gets there — 3 x 3: the one interior tile is the goal.  The oracle
    (and the engine: test_gpu_parity.py::test_sampler_give_up_status_bit) records it as the WARNING 1 in the agent's status word; the
    mirror's host draws (`rng` keyword) warn like the reference.  A 4 x 4 map has empty tiles: no warning."""
    import warnings

    import raycastworlds_jl_amd as RCW
    from raycastworlds_jl_amd import _capi

    orc = oracle.OracleBatch(4, seed=1, height_tile_map_tu=3, width_tile_map_tu=3, num_rays=8)
    assert (orc.status == _capi.RCW_WARN_SAMPLER_GAVE_UP).all() and (orc.goal == 2).all()
    frac = orc.position - np.floor(orc.position)
    assert (frac == 0.5).all()                                            # placed on a tile centre all the same (SR:125)
    orc.close()
    orc = oracle.OracleBatch(4, seed=1, height_tile_map_tu=4, width_tile_map_tu=4, num_rays=8)
    assert (orc.status == 0).all()
    orc.close()
    with pytest.warns(RuntimeWarning, match="Could not sample an empty position in max_tries = 9216"):
        gi, gj, ti, tj, d = RCW.SingleRoomModule.reference_reset_draws(np.random.default_rng(0), 3, 3, 128)
    assert (gi, gj) == (2, 2) and 1 <= ti <= 3 and 1 <= tj <= 3
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        RCW.SingleRoomModule.reference_reset_draws(np.random.default_rng(0), 4, 4, 128)
