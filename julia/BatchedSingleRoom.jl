# BatchedSingleRoom.jl — the reference-side binding of librcw_hip (include/rcw.h).
#
# This is what a RayCastWorlds.jl maintainer would add to keep the package's API
# (`RCW.reset!`, `RCW.act!`, `RCW.RLBaseEnv` with `state` / `reward` / `is_terminated`) while the
# SingleRoom step/render path runs on an MI355X for a whole batch of agents.
#
# NOT EXECUTED IN THIS PIPELINE: neither the build container nor the GPU box has a Julia
# toolchain (SURVEY.md §0), so this file is written against the C ABI and checked by reading
# only; the same calls are exercised by the Python/ctypes host layer in
# raycastworlds.jl_amd/ and by tests/.
#
# Usage (inside the RayCastWorlds module tree, next to single_room.jl):
#
#     include("BatchedSingleRoom.jl")
#     env = RCW.BatchedSingleRoomModule.BatchedSingleRoom(4096; height_tile_map_tu = 8,
#                                                         width_tile_map_tu = 8, num_rays = 256)
#     RCW.reset!(env)
#     RCW.act!(env, rand(UInt8(1):UInt8(4), 4096))
#     rl = RCW.RLBaseEnv(env)
#     obs = RLBase.state(rl)        # Array{UInt32,3}(H_cam, N, B) copied from the device
#     RLBase.reward(rl); RLBase.is_terminated(rl)

module BatchedSingleRoomModule

import ..RayCastWorlds as RCW
import ReinforcementLearningBase as RLBase

const librcw = get(ENV, "LIBRCW_HIP", "librcw_hip.so")

const NUM_ACTIONS = 4   # src/single_room.jl:19

# struct rcw_config (include/rcw.h) — field order and types must match exactly (160 bytes)
Base.@kwdef mutable struct RcwConfig
    abi_version::Int32 = 2
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
    write_columns::Int32 = 1
    out_of_bounds::Int32 = 0
    render_top_view::Int32 = 0
    world_unit_bits::Int32 = 32
    player_radius_wu_f64::Float64 = 1 / 8
    position_increment_wu_f64::Float64 = 1 / 8
    semi_field_of_view_wu_f64::Float64 = 2 / 3
    camera_height_tile_wu_f64::Float64 = 1
    reserved::NTuple{4, Int32} = (0, 0, 0, 0)
end

struct RcwError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:rcw_last_error, librcw), Cstring, ()))
    rc == -2 && throw(AssertionError(msg))            # @assert action in 1:4, single_room.jl:140
    rc == -5 && throw(BoundsError())                   # collision_detection.jl:35
    rc == -1 && throw(ArgumentError(msg))
    throw(RcwError(rc, msg))
end

mutable struct BatchedSingleRoom <: RCW.AbstractGame
    handle::Ptr{Cvoid}
    batch::Int
    config::RcwConfig
    camera_view::Array{UInt32, 3}     # (H_cam, N, B) host mirror, refreshed by state()
    seed::UInt64

    function BatchedSingleRoom(batch::Integer; device::Integer = 0, seed::Integer = 0, kwargs...)
        cfg = RcwConfig(; kwargs...)
        # Julia's own cos/sin for directions_wu (single_room.jl:65-69) can be handed over with
        # rcw_set_direction_table; the library's default is the same formula in C.
        handle = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:rcw_create, librcw), Cint, (Ref{RcwConfig}, Int32, Int32, UInt64, Ref{Ptr{Cvoid}}),
                    cfg, batch, device, seed, handle))
        view = Array{UInt32, 3}(undef, cfg.height_camera_view_pu, cfg.num_rays, batch)
        env = new(handle[], batch, cfg, view, seed)
        finalizer(e -> ccall((:rcw_destroy, librcw), Cint, (Ptr{Cvoid},), e.handle), env)
        return env
    end
end

# RCW.reset!(env)  — single_room.jl:326-331
function RCW.reset!(env::BatchedSingleRoom; mask::Union{Nothing, Vector{UInt8}} = nothing)
    check(ccall((:rcw_reset, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}, UInt64),
                env.handle, mask === nothing ? C_NULL : pointer(mask), env.seed))
    return nothing
end

# RCW.act!(env, action)  — single_room.jl:333-340; one action per agent (or one for all)
function RCW.act!(env::BatchedSingleRoom, actions::Vector{UInt8})
    length(actions) == env.batch || throw(DimensionMismatch("expected $(env.batch) actions"))
    check(ccall((:rcw_step, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}), env.handle, actions))
    return nothing
end
RCW.act!(env::BatchedSingleRoom, action::Integer) = RCW.act!(env, fill(UInt8(action), env.batch))

RCW.get_action_names(env::BatchedSingleRoom) = (:MOVE_FORWARD, :MOVE_BACKWARD, :TURN_LEFT, :TURN_RIGHT)

# Inject the state reset!(world) would have produced (single_room.jl:118-132) — how "identical
# seeds" is realised against the CPU reference (SURVEY.md §8c).
function set_state!(env::BatchedSingleRoom, goal_ij::Matrix{Int32}, position_wu::Matrix{Float32},
                    direction_au::Vector{Int32})
    check(ccall((:rcw_set_state, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Ptr{Int32}, Ptr{UInt8}),
                env.handle, goal_ij, position_wu, direction_au, C_NULL))
end

# world fields (single_room.jl:21-40)
function reward(env::BatchedSingleRoom)
    out = Vector{Float32}(undef, env.batch)
    check(ccall((:rcw_reward, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out)); out
end
function done(env::BatchedSingleRoom)
    out = Vector{UInt8}(undef, env.batch)
    check(ccall((:rcw_done, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt8}), env.handle, out)); out .!= 0
end
function player_position_wu(env::BatchedSingleRoom)
    out = Matrix{Float32}(undef, 2, env.batch)
    check(ccall((:rcw_position, librcw), Cint, (Ptr{Cvoid}, Ptr{Float32}), env.handle, out)); out
end
function player_direction_au(env::BatchedSingleRoom)
    out = Vector{Int32}(undef, env.batch)
    check(ccall((:rcw_direction, librcw), Cint, (Ptr{Cvoid}, Ptr{Int32}), env.handle, out)); out
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

# Device pointer of the observation batch (aliased, stable): for AMDGPU.jl users
#   unsafe_wrap(ROCArray{UInt32,3}, Ptr{UInt32}(ptr), (H_cam, N, B))
function camera_view_device_ptr(env::BatchedSingleRoom)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rcw_obs_device_ptr, librcw), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), env.handle, p)); p[]
end

#####
##### RLBase API  — single_room.jl:574-584
#####

RLBase.StateStyle(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = RLBase.Observation{Any}()
RLBase.state_space(env::RCW.RLBaseEnv{E}, ::RLBase.Observation) where {E <: BatchedSingleRoom} = nothing
function RLBase.state(env::RCW.RLBaseEnv{E}, ::RLBase.Observation) where {E <: BatchedSingleRoom}
    e = env.env
    check(ccall((:rcw_obs_copy, librcw), Cint, (Ptr{Cvoid}, Ptr{UInt32}, Int32, Int32),
                e.handle, e.camera_view, 0, e.batch))
    return e.camera_view            # the same Array every call, as in the reference (aliasing)
end
RLBase.reset!(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = RCW.reset!(env.env)
RLBase.action_space(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = Base.OneTo(NUM_ACTIONS)
(env::RCW.RLBaseEnv{E})(action) where {E <: BatchedSingleRoom} = RCW.act!(env.env, action)
RLBase.reward(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = reward(env.env)
RLBase.is_terminated(env::RCW.RLBaseEnv{E}) where {E <: BatchedSingleRoom} = done(env.env)

end # module
