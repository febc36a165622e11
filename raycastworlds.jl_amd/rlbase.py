"""`RLBaseEnv` wrapper (src/rlbase.jl:1-7) and the RLBase verbs the reference defines for
SingleRoom (src/single_room.jl:574-584), batched: one entry per agent."""
from __future__ import annotations

from . import single_room as _sr


class RLBaseEnv:
    """`struct RLBaseEnv{E} <: RLBase.AbstractEnv; env::E; end` (rlbase.jl:1-3)."""

    def __init__(self, env: "_sr.SingleRoom"):
        self.env = env

    def __call__(self, action):
        """`(env::RLBaseEnv)(action) = RCW.act!(env.env, action)` SR:581."""
        return _sr.act_(self.env, action)

    def __repr__(self):   # Base.show rlbase.jl:5
        e = self.env
        return (f"RLBaseEnv(SingleRoom(batch={e.batch}, {e.cfg.height_tile_map_tu}x{e.cfg.width_tile_map_tu}, "
                f"num_rays={e.cfg.num_rays}, device={e.device}))")


def state(env: RLBaseEnv):
    """`RLBase.state(env)` SR:576: `env.env.camera_view`, ALIASED device memory (no copy, no synchronisation);
    it is overwritten in place by the next action, as in the reference.  The same object until `bind_obs` moves it."""
    e = env.env
    cv = getattr(e, "_state_alias", None)
    if cv is None or cv.ptr != e._obs_ptr():
        cv = e._state_alias = e.camera_view
    return cv


def state_space(env: RLBaseEnv):
    """`RLBase.state_space` SR:575 returns nothing."""
    return None


def reset_(env: RLBaseEnv):
    """`RLBase.reset!(env)` SR:578."""
    return _sr.reset_(env.env)


def action_space(env: RLBaseEnv):
    """`RLBase.action_space(env) = Base.OneTo(NUM_ACTIONS)` SR:580."""
    return range(1, _sr.NUM_ACTIONS + 1)


def reward(env: RLBaseEnv):
    """`RLBase.reward(env) = env.env.world.reward` SR:583, where it lives: a `DeviceArray` R (B,) over the engine's reward
    array — the same object on every call, rewritten by every action in stream order, NO host synchronisation.  A
    GPU-resident agent takes `reward(env).torch(sync=False)`; host code uses it as an array (`reward(env) == 0`,
    `total += reward(env)`, `np.asarray(...)`), which copies it to the host at that moment."""
    return env.env.reward_device()


def is_terminated(env: RLBaseEnv):
    """`RLBase.is_terminated(env) = env.env.world.done` SR:584: Bool (B,), device-resident like `reward`."""
    return env.env.done_device(as_bool=True)
