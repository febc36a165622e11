# BatchedSingleRoom.jl — the reference-side binding of librcw_hip (include/rcw.h, ABI version 4).
#
# This is what a RayCastWorlds.jl maintainer would add to keep the package's API
# (`RCW.reset!`, `RCW.act!`, `RCW.cast_rays!`, `RCW.update_camera_view!`, `RCW.update_top_view!`,
# `RCW.RLBaseEnv` with `state` / `reward` / `is_terminated`) while the SingleRoom step/render path runs on an
# MI355X for a whole batch of agents.  EVERY export of include/rcw.h is bound here; tests/test_julia_binding.py
# checks each `ccall` below against the prototypes in the header (name, arity, pointer/scalar kind, width).
#
# NOT EXECUTED IN THIS PIPELINE: neither the build container nor the GPU box has a Julia toolchain
# (SURVEY.md §0), so this file is written against the C ABI and checked mechanically; the same calls are
# exercised by the Python/ctypes host layer in raycastworlds.jl_amd/ and by tests/c_abi_harness.c.
#
# Usage (inside the RayCastWorlds module tree, next to single_room.jl):
#
#     include("BatchedSingleRoom.jl")
#     env = RCW.BatchedSingleRoomModule.BatchedSingleRoom(4096; height_tile_map_tu = 8,
#                                                         width_tile_map_tu = 8, num_rays = 256)
#     RCW.reset!(env)
#     RCW.act!(env, rand(UInt8(1):UInt8(4), 4096))
#     rl = RCW.RLBaseEnv(env)
#     obs = RLBase.state(rl)        # the SAME Array{UInt32,3}(H_cam, N, B) every call (SR:576), refreshed lazily
#     RLBase.reward(rl); RLBase.is_terminated(rl)
#
# Observations.  The reference returns `camera_view` itself from `RLBase.state` (single_room.jl:576): one array,
# mutated in place by the next `act!`.  Here the frames live in HBM.  `RLBase.state` returns one host mirror —
# the same `Array` object on every call — and copies the batch off the device only when a step / reset has made
# the mirror stale, so calling it repeatedly costs nothing.  A device-side consumer skips the copy altogether:
# `camera_view_device_ptr(env)` is stable for the handle's lifetime, e.g. with AMDGPU.jl
# `unsafe_wrap(ROCArray{UInt32,3}, Ptr{UInt32}(ptr), (H_cam, N, B))`.

module BatchedSingleRoomModule

import ..RayCastWorlds as RCW
import MiniFB as MFB
import ReinforcementLearningBase as RLBase

const librcw = get(ENV, "LIBRCW_HIP", "librcw_hip.so")

const RCW_ABI_VERSION = 4
const NUM_ACTIONS = 4   # src/single_room.jl:19
const RCW_UNIQUE_ID_BYTES = 128
const RCW_GATHER_COLUMNS = Int32(0)
const RCW_GATHER_FRAMES = Int32(1)
# R of SingleRoom(; R = ...) single_room.jl:266  <->  rcw_config.reward_type
const REWARD_TYPES = (Float32, Float64, Int32, Int64)

# struct rcw_config (include/rcw.h) — field order and types must match exactly (160 bytes)
Base.@kwdef mutable struct RcwConfig
    abi_version::Int32 = RCW_ABI_VERSION
    height_tile_map_tu::Int32 = 8
    width_tile_map_tu::Int32 = 16
    num_directions::Int32 = 128
    num_rays::Int32 = 512
    height_camera_view_pu::Int32 = 256
    pu_per_tu::Int32 = 32
    player_radius_wu::Float32 = 1 / 8
    position_increment_wu::Float32 = 1 / 8
    semi_field_of_view_wu::Float32 = 2 / 3
    camera_height_tile_wu::Float32 = 1
    goal_reward::Float32 = 1
    floor_color::UInt32 = 0x00404040
    ceiling_color::UInt32 = 0x00FFFFFF
    wall_dim_1_color::UInt32 = 0x00808080
    wall_dim_2_color::UInt32 = 0x00c0c0c0
    goal_dim_1_color::UInt32 = 0x00800000
    goal_dim_2_color::UInt32 = 0x00c00000
    dda_tie_break::Int32 = 0
    dda_distance::Int32 = 0
    normalize_mode::Int32 = 0
    auto_reset::Int32 = 0
    agent_id_offset::Int64 = 0
    reward_type::Int32 = 0
    out_of_bounds::Int32 = 0
    render_top_view::Int32 = 0
    world_unit_bits::Int32 = 32
    player_radius_wu_f64::Float64 = 1 / 8
    position_increment_wu_f64::Float64 = 1 / 8
    semi_field_of_view_wu_f64::Float64 = 2 / 3
    camera_height_tile_wu_f64::Float64 = 1
    goal_reward_f64::Float64 = 1
    reserved::NTuple{2, Int32} = (0, 0)
end

struct RcwError <: Exception
    code::Cint
    msg::String
end

last_error() = unsafe_string(ccall((:rcw_last_error, librcw), Cstring, ()))
abi_version() = ccall((:rcw_abi_version, librcw), Cint, ())

function check(rc::Cint)
    rc == 0 && return nothing
    msg = last_error()
    rc == -2 && throw(AssertionError(msg))            # @assert action in 1:4, single_room.jl:140
    rc == -5 && throw(BoundsError())                   # collision_detection.jl:35
    rc == -1 && throw(ArgumentError(msg))
    rc == -4 && throw(OutOfMemoryError())
    throw(RcwError(rc, msg))
end

# The library's own defaults (the reference's kwargs, single_room.jl:258-272, 288-296): equals RcwConfig()
function config_default()
    cfg = RcwConfig()
    check(ccall((:rcw_config_default, librcw), Cint, (Ref{RcwConfig},), cfg))
    return cfg
end

mutable struct BatchedSingleRoom{T, R} <: RCW.AbstractGame
    handle::Ptr{Cvoid}
    batch::Int
    config::RcwConfig
    camera_view::Array{UInt32, 3}     # (H_cam, N, B) host mirror: THE array RLBase.state returns, every call
    stale::Bool                       # the mirror is older than the device frames
    seed::UInt64
    rng::Any                          # the reference's `rng` keyword (single_room.jl:49,265): nothing (resets are sampled on the
                                      # device, keyed by `seed`), an AbstractRNG, or a vector of them, one per agent
    gather_buffer::Ptr{Cvoid}         # device memory for gather_observations (allocated on first use)
    gather_frames::Int                # agents the gather buffer holds

    # SingleRoom(; T, R, kwargs...) single_room.jl:258-272 for `batch` agents on HIP device `device`
    function BatchedSingleRoom(batch::Integer; T::Type = Float32, R::Type = Float32, device::Integer = 0,
                               seed::Integer = 0, rng = nothing, kwargs...)
        T in (Float32, Float64) || throw(ArgumentError("T must be Float32 or Float64"))
        R in REWARD_TYPES || throw(ArgumentError("R must be one of $(REWARD_TYPES)"))
        cfg = RcwConfig(; kwargs...)
        # convert(T, .) of the caller's world-unit parameters (single_room.jl:263-270): for T = Float64 the library
        # reads the *_f64 fields — the Float64 value itself, not the Float32 one widened
        for name in (:player_radius_wu, :position_increment_wu, :semi_field_of_view_wu, :camera_height_tile_wu)
            haskey(kwargs, name) && setfield!(cfg, Symbol(name, :_f64), Float64(kwargs[name]))
        end
        cfg.world_unit_bits = T === Float64 ? 64 : 32
        cfg.reward_type = Int32(findfirst(==(R), REWARD_TYPES) - 1)
        cfg.goal_reward = one(Float32); cfg.goal_reward_f64 = 1.0      # one(R) single_room.jl:82
        handle = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:rcw_create, librcw), Cint, (Ref{RcwConfig}, Int32, Int32, UInt64, Ref{Ptr{Cvoid}}),
                    cfg, batch, device, seed, handle))
        view = Array{UInt32, 3}(undef, cfg.height_camera_view_pu, cfg.num_rays, batch)
        env = new{T, R}(handle[], batch, cfg, view, true, seed, rng, C_NULL, 0)
        finalizer(destroy!, env)
        # the reference's constructor consumes its rng twice: the draws of single_room.jl:62-74, then reset!(world) :105
        rng === nothing || reset_from_rng!(env, rng; construction = true)
        return env
    end
end

function destroy!(env::BatchedSingleRoom)
    env.handle == C_NULL && return nothing
    env.gather_buffer != C_NULL && ccall((:rcw_device_free, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), env.handle, env.gather_buffer)
    ccall((:rcw_destroy, librcw), Cint, (Ptr{Cvoid},), env.handle)
    env.handle = C_NULL
    env.gather_buffer = C_NULL
    return nothing
end

#####
##### the generic functions of RayCastWorlds.jl:7-14 on the path
#####

# reset!(world) single_room.jl:110-137 with the CALLER's generator — the reference's `rng` keyword (single_room.jl:49,265).
# The draws are made on the host by the reference's own statements, in its order: `rand(rng, 2:H-1)`, `rand(rng, 2:W-1)`
# (:120), `RCW.sample_empty_position(rng, tile_map)` (:124 — the package's own function on a host tile map with the wall ring
# and the new goal, utils.jl:23-58), `rand(rng, 0:nd-1)` (:128) — agent after agent from one generator (B reference worlds
# sharing it, reset in order), or agent a from `rng[a]` (each agent IS the reference world built with that generator: its
# stream is the reference's, draw for draw).  The state goes to the engine with one rcw_set_state.
function reset_from_rng!(env::BatchedSingleRoom{T}, rng; mask::Union{Nothing, Vector{UInt8}} = nothing,
                         construction::Bool = false) where {T}
    H, W, nd, B = Int(env.config.height_tile_map_tu), Int(env.config.width_tile_map_tu), Int(env.config.num_directions), env.batch
    rng isa AbstractVector && length(rng) != B && throw(DimensionMismatch("expected one generator or $(B) of them"))
    goal = fill(Int32(2), 2, B); position = fill(T(1.5), 2, B); direction = zeros(Int32, B)
    tile_map = falses(2, H, W)                                                   # NUM_OBJECTS = 2, WALL = 1, GOAL = 2 (:16-18)
    tile_map[1, :, 1] .= true; tile_map[1, :, W] .= true; tile_map[1, 1, :] .= true; tile_map[1, H, :] .= true   # :57-60
    for a in 1:B
        (mask !== nothing && mask[a] == 0) && continue
        g = rng isa AbstractVector ? rng[a] : rng
        for pass in (construction ? (1, 2) : (2,))                              # (the constructor's own draws :62-74 come first)
            goal_position = CartesianIndex(rand(g, 2 : H - 1), rand(g, 2 : W - 1))                    # :62 / :120
            tile_map[2, goal_position] = true                                                         # :63 / :122
            player_position_tu = RCW.sample_empty_position(g, tile_map)                               # :71 / :124
            player_direction_au = rand(g, 0 : nd - 1)                                                 # :74 / :128
            tile_map[2, goal_position] = false                                                        # (:118 of the next reset!)
            goal[1, a] = goal_position[1]; goal[2, a] = goal_position[2]
            position[1, a] = convert(T, player_position_tu[1] - 0.5); position[2, a] = convert(T, player_position_tu[2] - 0.5)   # :125
            direction[a] = player_direction_au
        end
    end
    set_state!(env, goal, position, direction; mask = mask)
    return nothing
end

# RCW.reset!(env)  — single_room.jl:326-331 (all agents, or those whose mask byte is non-zero).  With `rng` (or an
# environment built with one): the caller's generator, as the reference; else sampled on the device, keyed by `seed`.
function RCW.reset!(env::BatchedSingleRoom; mask::Union{Nothing, Vector{UInt8}} = nothing, rng = env.rng,
                    seed::Union{Nothing, Integer} = nothing)
    if rng !== nothing && seed === nothing
        return reset_from_rng!(env, rng; mask = mask)
    end
    env.seed = seed === nothing ? env.seed : seed
    GC.@preserve mask check(ccall((:rcw_reset, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}, UInt64),
                                  env.handle, mask === nothing ? Ptr{UInt8}(C_NULL) : pointer(mask), env.seed))
    env.stale = true
    return nothing
end

# RCW.act!(env, action)  — single_room.jl:333-340; one action per agent (or one for all)
function RCW.act!(env::BatchedSingleRoom, actions::Vector{UInt8})
    length(actions) == env.batch || throw(DimensionMismatch("expected $(env.batch) actions"))
    check(ccall((:rcw_step, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}), env.handle, actions))
    env.stale = true
    return nothing
end
RCW.act!(env::BatchedSingleRoom, action::Integer) = RCW.act!(env, fill(UInt8(action), env.batch))

# The same with actions already in device memory (a policy that runs on the GPU): stream-ordered, no host round trip
function act_device!(env::BatchedSingleRoom, actions_device::Ptr{UInt8})
    check(ccall((:rcw_step_device, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}), env.handle, actions_device))
    env.stale = true
    return nothing
end

RCW.get_action_keys(env::BatchedSingleRoom) = (MFB.KB_KEY_W, MFB.KB_KEY_S, MFB.KB_KEY_A, MFB.KB_KEY_D)   # single_room.jl:485
RCW.get_action_names(env::BatchedSingleRoom) = (:MOVE_FORWARD, :MOVE_BACKWARD, :TURN_LEFT, :TURN_RIGHT)   # single_room.jl:486

# RCW.cast_rays!(world) single_room.jl:195-231, RCW.update_camera_view!(env) :374-444, RCW.update_top_view!(env) :446-483
function RCW.cast_rays!(env::BatchedSingleRoom)
    check(ccall((:rcw_cast_rays, librcw), Cint, (Ptr{Cvoid},), env.handle))
    return nothing
end
function RCW.update_camera_view!(env::BatchedSingleRoom)
    check(ccall((:rcw_update_camera_view, librcw), Cint, (Ptr{Cvoid},), env.handle))
    env.stale = true
    return nothing
end
function RCW.update_top_view!(env::BatchedSingleRoom)
    check(ccall((:rcw_update_top_view, librcw), Cint, (Ptr{Cvoid},), env.handle))
    return nothing
end

sync(env::BatchedSingleRoom) = check(ccall((:rcw_sync, librcw), Cint, (Ptr{Cvoid},), env.handle))
clear_error!(env::BatchedSingleRoom) = check(ccall((:rcw_clear_error, librcw), Cint, (Ptr{Cvoid},), env.handle))

# Inject the state reset!(world) would have produced (single_room.jl:118-132) — how "identical
# seeds" is realised against the CPU reference (SURVEY.md §8c).
function set_state!(env::BatchedSingleRoom{Float32}, goal_ij::Matrix{Int32}, position_wu::Matrix{Float32},
                    direction_au::Vector{Int32}; mask::Union{Nothing, Vector{UInt8}} = nothing)
    GC.@preserve mask check(ccall((:rcw_set_state, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Ptr{Int32}, Ptr{UInt8}),
                                  env.handle, goal_ij, position_wu, direction_au,
                                  mask === nothing ? Ptr{UInt8}(C_NULL) : pointer(mask)))
    env.stale = true
    return nothing
end
function set_state!(env::BatchedSingleRoom{Float64}, goal_ij::Matrix{Int32}, position_wu::Matrix{Float64},
                    direction_au::Vector{Int32}; mask::Union{Nothing, Vector{UInt8}} = nothing)
    GC.@preserve mask check(ccall((:rcw_set_state64, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}, Ptr{Int32}, Ptr{UInt8}),
                                  env.handle, goal_ij, position_wu, direction_au,
                                  mask === nothing ? Ptr{UInt8}(C_NULL) : pointer(mask)))
    env.stale = true
    return nothing
end

# Julia's own directions_wu (single_room.jl:65-69: Julia's cos / sin) instead of the C library's
function set_direction_table!(env::BatchedSingleRoom{Float32}, directions_wu::Matrix{Float32})
    check(ccall((:rcw_set_direction_table, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, directions_wu))
    env.stale = true
end
function set_direction_table!(env::BatchedSingleRoom{Float64}, directions_wu::Matrix{Float64})
    check(ccall((:rcw_set_direction_table64, librcw), Cint, (Ptr{Cvoid}, Ptr{Float64}), env.handle, directions_wu))
    env.stale = true
end
function julia_direction_table(::Type{T}, num_directions) where {T}
    out = Matrix{T}(undef, 2, num_directions)
    for i in 1:num_directions
        theta_wu = (i - 1) * 2 * pi / num_directions                     # single_room.jl:66
        out[1, i] = convert(T, cos(theta_wu)); out[2, i] = convert(T, sin(theta_wu))
    end
    return out
end

#####
##### world fields (single_room.jl:21-40)
#####

function reward(env::BatchedSingleRoom{T, R}) where {T, R}
    out = Vector{R}(undef, env.batch)
    check(ccall((:rcw_reward_typed, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), env.handle, out)); out
end
function reward_float32(env::BatchedSingleRoom{T, Float32}) where {T}
    out = Vector{Float32}(undef, env.batch)
    check(ccall((:rcw_reward, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out)); out
end
function done(env::BatchedSingleRoom)
    out = Vector{UInt8}(undef, env.batch)
    check(ccall((:rcw_done, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}), env.handle, out)); out .!= 0
end
function player_position_wu(env::BatchedSingleRoom{Float32})
    out = Matrix{Float32}(undef, 2, env.batch)
    check(ccall((:rcw_position, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out)); out
end
function player_position_wu(env::BatchedSingleRoom{Float64})
    out = Matrix{Float64}(undef, 2, env.batch)
    check(ccall((:rcw_position64, librcw), Cint, (Ptr{Cvoid}, Ptr{Float64}), env.handle, out)); out
end
function player_direction_au(env::BatchedSingleRoom)
    out = Vector{Int32}(undef, env.batch)
    check(ccall((:rcw_direction, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}), env.handle, out)); out
end
function goal_position(env::BatchedSingleRoom)
    out = Matrix{Int32}(undef, 2, env.batch)
    check(ccall((:rcw_goal, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}), env.handle, out))
    return [CartesianIndex(Int(out[1, b]), Int(out[2, b])) for b in 1:env.batch]
end
function episode(env::BatchedSingleRoom)
    out = Vector{UInt32}(undef, env.batch)
    check(ccall((:rcw_episode, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt32}), env.handle, out)); out
end
# per-agent sticky status: 0, -5 where the reference would have raised BoundsError, -2 for an invalid device action,
# 1 (a warning) where sample_empty_position gave up after max_tries and returned an occupied tile (utils.jl:34 @warns there)
function status(env::BatchedSingleRoom)
    out = Vector{Int32}(undef, env.batch)
    check(ccall((:rcw_status, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}), env.handle, out)); out
end
# tile_map as B BitArray{3}(2, H, W): the library hands back `.chunks` verbatim
function tile_maps(env::BatchedSingleRoom)
    n = Ref{Int32}(0)
    check(ccall((:rcw_tile_map_num_chunks, librcw), Cint, (Ptr{Cvoid}, Ref{Int32}), env.handle, n))
    chunks = Matrix{UInt64}(undef, n[], env.batch)
    check(ccall((:rcw_tile_map_chunks, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt64}), env.handle, chunks))
    H, W = env.config.height_tile_map_tu, env.config.width_tile_map_tu
    return map(1:env.batch) do b
        tm = falses(2, H, W)
        copyto!(tm.chunks, view(chunks, :, b))
        tm
    end
end

# world.ray_stop_position_tu / ray_hit_dimension / ray_distance_wu / ray_directions_wu (single_room.jl:29-31,39)
# of agents first+1 : first+count (0-based `first`, as in the C ABI)
function rays(env::BatchedSingleRoom{Float32}; first::Integer = 0, count::Integer = env.batch - first)
    N = Int(env.config.num_rays)
    stop = Array{Int64, 3}(undef, 2, N, count); dim = Matrix{Int64}(undef, N, count)
    dist = Matrix{Float32}(undef, N, count); dirs = Array{Float32, 3}(undef, 2, N, count)
    check(ccall((:rcw_rays, librcw), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float32}, Ptr{Float32}),
                env.handle, first, count, stop, dim, dist, dirs))
    return (ray_stop_position_tu = stop, ray_hit_dimension = dim, ray_distance_wu = dist, ray_directions_wu = dirs)
end
function rays(env::BatchedSingleRoom{Float64}; first::Integer = 0, count::Integer = env.batch - first)
    N = Int(env.config.num_rays)
    stop = Array{Int64, 3}(undef, 2, N, count); dim = Matrix{Int64}(undef, N, count)
    dist = Matrix{Float64}(undef, N, count); dirs = Array{Float64, 3}(undef, 2, N, count)
    check(ccall((:rcw_rays64, librcw), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
                env.handle, first, count, stop, dim, dist, dirs))
    return (ray_stop_position_tu = stop, ray_hit_dimension = dim, ray_distance_wu = dist, ray_directions_wu = dirs)
end

# directions_wu (single_room.jl:28) and the (direction, ray) table the kernels use: (N, 5, nd)
function directions_wu(env::BatchedSingleRoom{Float32})
    out = Matrix{Float32}(undef, 2, env.config.num_directions)
    check(ccall((:rcw_direction_table, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out)); out
end
function directions_wu(env::BatchedSingleRoom{Float64})
    out = Matrix{Float64}(undef, 2, env.config.num_directions)
    check(ccall((:rcw_direction_table64, librcw), Cint, (Ptr{Cvoid}, Ptr{Float64}), env.handle, out)); out
end
function ray_table(env::BatchedSingleRoom{Float32})
    out = Array{Float32, 3}(undef, env.config.num_rays, 5, env.config.num_directions)
    check(ccall((:rcw_ray_table, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out)); out
end
function ray_table(env::BatchedSingleRoom{Float64})
    out = Array{Float64, 3}(undef, env.config.num_rays, 5, env.config.num_directions)
    check(ccall((:rcw_ray_table64, librcw), Cint, (Ptr{Cvoid}, Ptr{Float64}), env.handle, out)); out
end

#####
##### images
#####

# Device pointer of the observation batch (aliased, stable): for AMDGPU.jl users
#   unsafe_wrap(ROCArray{UInt32,3}, Ptr{UInt32}(ptr), (H_cam, N, B))
function camera_view_device_ptr(env::BatchedSingleRoom)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_obs_device_ptr, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), env.handle, p)); p[]
end
# The host mirror, refreshed only if a step / reset happened since the last refresh (the SAME array every call)
function camera_view(env::BatchedSingleRoom)
    if env.stale
        check(ccall((:rcw_obs_copy, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Int32, Int32),
                    env.handle, env.camera_view, 0, env.batch))
        env.stale = false
    end
    return env.camera_view
end
# Render into caller-owned device memory instead (double-buffered observations); C_NULL restores the library's buffer
bind_obs!(env::BatchedSingleRoom, device_ptr::Ptr{Cvoid}) =
    check(ccall((:rcw_bind_obs, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), env.handle, device_ptr))

# env.top_view (single_room.jl:302; build with render_top_view = 1): (H*pu, W*pu, count) copied to host
function top_view(env::BatchedSingleRoom; first::Integer = 0, count::Integer = env.batch - first)
    c = env.config
    out = Array{UInt32, 3}(undef, c.height_tile_map_tu * c.pu_per_tu, c.width_tile_map_tu * c.pu_per_tu, count)
    check(ccall((:rcw_top_view_copy, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Int32, Int32), env.handle, out, first, count)); out
end
function top_view_device_ptr(env::BatchedSingleRoom)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_top_view_device_ptr, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), env.handle, p)); p[]
end
function reward_device_ptr(env::BatchedSingleRoom)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_reward_device_ptr, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), env.handle, p)); p[]
end
function done_device_ptr(env::BatchedSingleRoom)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_done_device_ptr, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), env.handle, p)); p[]
end

# The RLBase verbs for a GPU-RESIDENT agent: where `RLBase.state / reward / is_terminated` below hand out host arrays
# (one device-to-host copy and a wait for the engine's stream per call), these hand out what the engine itself writes —
# (device pointer, element type, dims), stable for the handle's lifetime and refreshed by every step in stream order, to
# be wrapped once, e.g. with AMDGPU.jl `unsafe_wrap(ROCArray{R,1}, Ptr{R}(ptr), dims)`.  A loop
# `act!(env, actions_device_ptr) -> policy kernels on stream(env)` then never synchronises the host (the Python mirror's
# RLBase.reward / is_terminated return exactly these aliases by default; bench.py --api rlbase measures the loop).
state_device(env::BatchedSingleRoom) =
    (ptr = camera_view_device_ptr(env), eltype = UInt32,
     dims = (Int(env.config.height_camera_view_pu), Int(env.config.num_rays), env.batch))
reward_device(env::BatchedSingleRoom{T, R}) where {T, R} = (ptr = reward_device_ptr(env), eltype = R, dims = (env.batch,))
is_terminated_device(env::BatchedSingleRoom) = (ptr = done_device_ptr(env), eltype = UInt8, dims = (env.batch,))

# The compact per-column descriptor of the frames: height_line_pu (single_room.jl:408-411) and colour id by image column
function columns(env::BatchedSingleRoom; first::Integer = 0, count::Integer = env.batch - first)
    N = Int(env.config.num_rays)
    h = Matrix{Int32}(undef, N, count); c = Matrix{UInt8}(undef, N, count)
    check(ccall((:rcw_columns, librcw), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{UInt8}), env.handle, first, count, h, c))
    return (height_line_pu = h, colour_id = c)
end
function columns_device_ptr(env::BatchedSingleRoom)
    h = Ref{Ptr{Cvoid}}(C_NULL); c = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_columns_device_ptr, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}), env.handle, h, c))
    return (h[], c[])
end
# descriptors (device pointers) -> frames (device pointer), with this handle's colours: the receiving side of a gather
expand_columns!(env::BatchedSingleRoom, height_line_pu_device::Ptr{Int32}, colour_id_device::Ptr{UInt8}, count::Integer,
                frames_device::Ptr{Cvoid}) =
    check(ccall((:rcw_expand_columns, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{UInt8}, Int32, Ptr{Cvoid}),
                env.handle, height_line_pu_device, colour_id_device, count, frames_device))

#####
##### streams, device memory, timing, introspection
#####

set_stream!(env::BatchedSingleRoom, hip_stream::Ptr{Cvoid}) =
    check(ccall((:rcw_set_stream, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), env.handle, hip_stream))
function get_stream(env::BatchedSingleRoom)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_get_stream, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), env.handle, p)); p[]
end
function device_malloc(env::BatchedSingleRoom, bytes::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_device_malloc, librcw), Cint, (Ptr{Cvoid}, UInt64, Ref{Ptr{Cvoid}}), env.handle, bytes, p)); p[]
end
device_free(env::BatchedSingleRoom, p::Ptr{Cvoid}) =
    check(ccall((:rcw_device_free, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), env.handle, p))
memcpy_to_host!(env::BatchedSingleRoom, dst::Array, src_device::Ptr{Cvoid}) =
    check(ccall((:rcw_memcpy_to_host, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt64),
                env.handle, dst, src_device, sizeof(dst)))

timer_start(env::BatchedSingleRoom) = check(ccall((:rcw_timer_start, librcw), Cint, (Ptr{Cvoid},), env.handle))
function timer_stop(env::BatchedSingleRoom)
    ms = Ref{Float32}(0)
    check(ccall((:rcw_timer_stop, librcw), Cint, (Ptr{Cvoid}, Ref{Float32}), env.handle, ms)); ms[]
end
profile!(env::BatchedSingleRoom, enable::Bool) =
    check(ccall((:rcw_profile, librcw), Cint, (Ptr{Cvoid}, Int32), env.handle, enable ? 1 : 0))
function profile_read(env::BatchedSingleRoom)
    c = Ref{Float32}(0); t = Ref{Float32}(0); f = Ref{Float32}(0); n = Ref{Int32}(0)
    check(ccall((:rcw_profile_read, librcw), Cint, (Ptr{Cvoid}, Ref{Float32}, Ref{Float32}, Ref{Float32}, Ref{Int32}),
                env.handle, c, t, f, n))
    return (cast_ms = c[], top_view_ms = t[], fill_ms = f[], steps = n[])
end
"Which kernel form `update_top_view!` takes: `:none`, `:in_place`, `:one_kernel` or `:two_kernels` (rcw.h)."
function top_view_form(env::BatchedSingleRoom)
    f = Ref{Int32}(0)
    check(ccall((:rcw_top_view_form, librcw), Cint, (Ptr{Cvoid}, Ref{Int32}), env.handle, f))
    return (:none, :in_place, :one_kernel, :two_kernels)[f[] + 1]
end
"The form `update_top_view!(env)` takes when it is called alone, outside a step (`rcw_update_top_view_form`)."
function update_top_view_form(env::BatchedSingleRoom)
    f = Ref{Int32}(0)
    check(ccall((:rcw_update_top_view_form, librcw), Cint, (Ptr{Cvoid}, Ref{Int32}), env.handle, f))
    return (:none, :in_place, :one_kernel, :two_kernels)[f[] + 1]
end
"""
    set_top_view_form!(env, form = :auto; runs = 0)

Choose the kernel form of `update_top_view!` instead of the library's rule (`rcw_set_top_view_form`): `:auto`,
`:in_place`, `:one_kernel` or `:two_kernels`; `runs` = 0 (automatic) or 1..8 runs of agents for the two-kernel form.
All forms write the same pixels.  Throws when the geometry cannot take the form (the handle keeps the automatic one).
"""
function set_top_view_form!(env::BatchedSingleRoom, form::Symbol = :auto; runs::Integer = 0)
    code = Dict(:auto => 0, :in_place => 1, :one_kernel => 2, :two_kernels => 3)[form]
    check(ccall((:rcw_set_top_view_form, librcw), Cint, (Ptr{Cvoid}, Int32, Int32), env.handle, code, runs))
    return nothing
end
"How many launches a step takes (`rcw_step_form`): `:two_launches` (cast kernel, then fill kernel) or `:one_launch`."
function step_form(env::BatchedSingleRoom)
    f = Ref{Int32}(0)
    check(ccall((:rcw_step_form, librcw), Cint, (Ptr{Cvoid}, Ref{Int32}), env.handle, f))
    return (:two_launches, :one_launch)[f[]]
end
"""
    set_step_form!(env, form = :auto)

Choose the step's form instead of the library's rule (`rcw_set_step_form`): `:auto`, `:two_launches` or `:one_launch`
(the fill workgroups of a launch write the frames the actions select among the successor states the previous launch cast).
Both leave the same state and the same pixels.  Throws where the geometry cannot take the one-launch form.
"""
function set_step_form!(env::BatchedSingleRoom, form::Symbol = :auto)
    code = Dict(:auto => 0, :two_launches => 1, :one_launch => 2)[form]
    check(ccall((:rcw_set_step_form, librcw), Cint, (Ptr{Cvoid}, Int32), env.handle, code))
    return nothing
end
"The kernel `update_camera_view!` takes for this camera height and batch (`rcw_fill_kernel_name`)."
function fill_kernel_name(env::BatchedSingleRoom)
    buf = Vector{UInt8}(undef, 64)
    check(ccall((:rcw_fill_kernel_name, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32), env.handle, buf, length(buf)))
    return unsafe_string(pointer(buf))
end
function batch(env::BatchedSingleRoom)
    n = Ref{Int32}(0)
    check(ccall((:rcw_batch, librcw), Cint, (Ptr{Cvoid}, Ref{Int32}), env.handle, n)); Int(n[])
end
function get_config(env::BatchedSingleRoom)
    cfg = RcwConfig()
    check(ccall((:rcw_get_config, librcw), Cint, (Ptr{Cvoid}, Ref{RcwConfig}), env.handle, cfg)); cfg
end
function device_name(env::BatchedSingleRoom)
    buf = Vector{UInt8}(undef, 256)
    check(ccall((:rcw_device_name, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32), env.handle, buf, length(buf)))
    return unsafe_string(pointer(buf))
end

#####
##### multi-GPU: one process (or task) per GPU, agents sharded by rank; the observation gather over RCCL
#####
# rank 0 makes the id and ships its 128 bytes to the other ranks (Distributed.jl, MPI.jl, a file — caller's choice)
function comm_unique_id()
    id = Vector{UInt8}(undef, RCW_UNIQUE_ID_BYTES)
    check(ccall((:rcw_comm_unique_id, librcw), Cint, (Ptr{Cvoid},), id)); id
end
comm_init!(env::BatchedSingleRoom, unique_id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(ccall((:rcw_comm_init, librcw), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32), env.handle, unique_id, rank, world))
comm_destroy!(env::BatchedSingleRoom) = check(ccall((:rcw_comm_destroy, librcw), Cint, (Ptr{Cvoid},), env.handle))
function comm_info(env::BatchedSingleRoom)
    r = Ref{Int32}(0); w = Ref{Int32}(0)
    check(ccall((:rcw_comm_info, librcw), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), env.handle, r, w))
    return (rank = Int(r[]), world = Int(w[]))
end
# all-gather of the compact descriptors into caller-owned device memory (B*world columns each)
gather_columns!(env::BatchedSingleRoom, height_line_pu_all_device::Ptr{Int32}, colour_id_all_device::Ptr{UInt8}) =
    check(ccall((:rcw_gather_columns, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{UInt8}),
                env.handle, height_line_pu_all_device, colour_id_all_device))
# the GLOBAL observation batch (H_cam, N, B*world) into caller-owned device memory
gather_observations!(env::BatchedSingleRoom, frames_all_device::Ptr{Cvoid}; mode::Int32 = RCW_GATHER_COLUMNS) =
    check(ccall((:rcw_gather_observations, librcw), Cint, (Ptr{Cvoid}, Int32, Ptr{Cvoid}), env.handle, mode, frames_all_device))
# ... and as a host Array, through a device buffer the binding keeps (what a learner on rank 0 would read)
function gather_observations(env::BatchedSingleRoom; mode::Int32 = RCW_GATHER_COLUMNS)
    world = comm_info(env).world
    world >= 1 || throw(ArgumentError("comm_init! first"))
    c = env.config
    out = Array{UInt32, 3}(undef, c.height_camera_view_pu, c.num_rays, env.batch * world)
    if env.gather_frames != env.batch * world
        env.gather_buffer != C_NULL && device_free(env, env.gather_buffer)
        env.gather_buffer = device_malloc(env, sizeof(out))
        env.gather_frames = env.batch * world
    end
    gather_observations!(env, env.gather_buffer; mode = mode)
    memcpy_to_host!(env, out, env.gather_buffer)
    return out
end

#####
##### RLBase API  — single_room.jl:574-584
#####

RLBase.StateStyle(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = RLBase.Observation{Any}()
RLBase.state_space(env::RCW.RLBaseEnv{E}, ::RLBase.Observation) where {E <: BatchedSingleRoom} = nothing
RLBase.state(env::RCW.RLBaseEnv{E}, ::RLBase.Observation) where {E <: BatchedSingleRoom} = camera_view(env.env)
RLBase.reset!(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = RCW.reset!(env.env)
RLBase.action_space(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = Base.OneTo(NUM_ACTIONS)
(env::RCW.RLBaseEnv{E})(action) where {E <: BatchedSingleRoom} = RCW.act!(env.env, action)
RLBase.reward(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = reward(env.env)
RLBase.is_terminated(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = done(env.env)

end # module
