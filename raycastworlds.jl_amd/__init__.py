"""raycastworlds.jl_amd — MI355X-native batched SingleRoom step/render engine.

A drop-in for ONE path of RayCastWorlds.jl (SingleRoom `reset!` / `act!` / camera view):
hand-written HIP kernels for gfx950 behind a C ABI (include/rcw.h), with this package as
the host-side mirror of the reference's interface:

    import raycastworlds_jl_amd as RCW
    env = RCW.SingleRoomModule.SingleRoom(batch=4096, height_tile_map_tu=8,
                                          width_tile_map_tu=8, num_rays=256)
    RCW.reset_(env)                       # RCW.reset!(env)
    RCW.act_(env, actions)                # RCW.act!(env, action), one action per agent
    rl = RCW.RLBaseEnv(env)
    RCW.RLBase.state(rl); RCW.RLBase.reward(rl); RCW.RLBase.is_terminated(rl)

The directory name contains a dot, so import it through the `raycastworlds_jl_amd` shim
at the repository root.
"""
from . import rlbase as RLBase
from . import single_room as SingleRoomModule
from .rlbase import RLBaseEnv
from .sharded import ShardedSingleRoom, make_unique_id, shard_range
from .viewer import frame_buffer_of, frame_to_rgb, play_keys, save_agent_ppm, save_ppm
from .single_room import (act_, cast_rays_, get_action_keys, get_action_names, pu_to_tu, reset_, update_camera_view_, update_top_view_, wu_to_pu, wu_to_tu)

__all__ = ["SingleRoomModule", "RLBase", "RLBaseEnv", "ShardedSingleRoom", "shard_range", "make_unique_id", "frame_to_rgb", "save_ppm", "save_agent_ppm", "play_keys", "frame_buffer_of", "reset_", "act_", "cast_rays_",
           "update_camera_view_", "update_top_view_", "get_action_names", "get_action_keys", "wu_to_tu", "wu_to_pu", "pu_to_tu"]
