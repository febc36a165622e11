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
