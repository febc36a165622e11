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
    """`RLBase.state(env)` SR:576: `env.env.camera_view`, ALIASED device memory (no copy);
    it is overwritten in place by the next action, as in the reference."""
    return env.env.camera_view


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
    """`RLBase.reward(env)` SR:583: Float32 (B,)."""
    return env.env.world.reward


def is_terminated(env: RLBaseEnv):
    """`RLBase.is_terminated(env)` SR:584: Bool (B,)."""
    return env.env.world.done
