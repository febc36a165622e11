#!/usr/bin/env julia
# make_reference_fixtures.jl — ONE command that turns "parity unpinned" into a verdict.
#
#     julia --project=<environment with RayCastWorlds 0.1.x installed> julia/make_reference_fixtures.jl [outdir]
#
# Runs the REAL RayCastWorlds.jl (with its real RayCaster / StaticArrays / SimpleDraw dependencies) on the
# discriminating inputs of julia/discriminator_cases.tsv and writes, per case, exactly what the reference
# computes for that injected state:
#
#     <outdir>/<case>.txt                 directions_wu, ray_directions_wu, ray_stop_position_tu, ray_hit_dimension,
#                                         ray_distance_wu (floats as IEEE-754 bit patterns), a 64-step rollout trace
#     <outdir>/<case>.camera_view.u32     env.camera_view, raw little-endian UInt32, Julia column-major (H_cam, N)
#     <outdir>/<case>.top_view.u32        env.top_view, raw (H*pu, W*pu) — pins SimpleDraw's Line / Circle too
#     <outdir>/seeded_rng_mt1_resets.txt  the `rng` keyword (single_room.jl:49,265): SingleRoom(; rng = MersenneTwister(1)), a 64-step
#                                         rollout with two reset!(env) in it; the state after construction and after every reset!
#                                         is dumped, so a consumer without Julia's generator can replay the trajectory
#     <outdir>/manifest.tsv               the cases written + package versions
#
# Default outdir: tests/golden/reference (next to this repository's other fixtures).  Then
#
#     python -m pytest tests/test_reference_fixtures.py            # CPU: decides every UNPINNED switch
#     python -m pytest tests/test_reference_fixtures.py -m gpu     # the HIP path against the same files
#
# report which setting of dda_tie_break / dda_distance / normalize_mode (include/rcw.h) reproduces the reference
# bit for bit, whether Julia's cos/sin table equals the C one, and whether the top-view rasterisers match.
#
# NOT EXECUTED IN THIS PIPELINE (no Julia toolchain on either box, SURVEY.md §0).  It uses only Base and the
# package itself: no JSON / NPZ / DelimitedFiles dependency.
#
# How a state is injected (what the C ABI's rcw_set_state does, include/rcw.h): the fields reset!(world) writes at
# src/single_room.jl:118-132 are assigned directly, then cast_rays! (SR:134) and the two renderers (SR:328-329) run.

import RayCastWorlds as RCW
import Random

const SRM = RCW.SingleRoomModule
const HERE = @__DIR__
const OUT = length(ARGS) >= 1 ? ARGS[1] : joinpath(HERE, "..", "tests", "golden", "reference")

bits(x::Float32) = Int(reinterpret(UInt32, x))
bits(x::Float64) = reinterpret(UInt64, x)
f32(b::AbstractString) = reinterpret(Float32, parse(UInt32, b))

function inject!(env, goal_i, goal_j, x, y, direction_au)
    world = env.world
    world.tile_map[SRM.GOAL, world.goal_position] = false                      # SR:118
    world.goal_position = CartesianIndex(goal_i, goal_j)                       # SR:121
    world.tile_map[SRM.GOAL, world.goal_position] = true                       # SR:122
    world.player_position_wu = typeof(world.player_position_wu)(x, y)          # SR:126
    world.player_direction_au = direction_au                                   # SR:129
    world.reward = zero(world.reward)                                          # SR:131
    world.done = false                                                         # SR:132
    RCW.cast_rays!(world)                                                      # SR:134
    RCW.update_top_view!(env)                                                  # SR:328
    RCW.update_camera_view!(env)                                               # SR:329
    return nothing
end

# the action stream of tests/c_abi_harness.c (a 64-bit LCG), so every consumer can regenerate it
function lcg_actions(n, seed::UInt64)
    s = seed
    out = Vector{Int}(undef, n)
    for k in 1:n
        s = s * 0x5851f42d4c957f2d + 0x14057b7ef767814f
        out[k] = 1 + Int((s >> 33) % 4)
    end
    return out
end

join_ints(v) = join(string.(v), " ")

function pkg_version(m)
    try
        return string(Base.pkgversion(m))          # Julia >= 1.9
    catch
        return "unknown"
    end
end

# The seeded case: the reference's own `rng` keyword.  The post-reset states are written out (goal tile, position bits,
# heading), because no consumer outside Julia can reproduce MersenneTwister's stream: tests/reference_pin.py injects them
# (rcw_set_state) where this script calls reset!(env), and everything between two resets must then agree step for step —
# which also pins that a reset leaves reward = 0, done = false and freshly cast rays (single_room.jl:131-134).
const SEEDED_NAME = "seeded_rng_mt1_resets"
const SEEDED_RESET_STEPS = (20, 45)              # reset!(env) is called BEFORE the action of these (1-based) steps
function seeded_case(versions)
    rng = Random.MersenneTwister(1)
    H, W, N = 8, 8, 64
    env = SRM.SingleRoom(height_tile_map_tu = H, width_tile_map_tu = W, num_rays = N, rng = rng)
    world = env.world
    state() = (world.goal_position[1], world.goal_position[2], bits(world.player_position_wu[1]), bits(world.player_position_wu[2]),
               world.player_direction_au)
    states = Int[]; append!(states, state())                                     # after construction (rng consumed twice)
    write(joinpath(OUT, SEEDED_NAME * ".camera_view.u32"), env.camera_view)
    write(joinpath(OUT, SEEDED_NAME * ".top_view.u32"), env.top_view)
    actions = lcg_actions(64, UInt64(7))
    pos, dirs, rew, done, goals = Int[], Int[], Int[], Int[], Int[]
    error_step = 0
    for (k, a) in enumerate(actions)
        if k in SEEDED_RESET_STEPS
            RCW.reset!(env)                                                      # single_room.jl:326-331, draws from rng
            append!(states, state())
        end
        try
            RCW.act!(env, a)
        catch err
            err isa BoundsError || rethrow()
            error_step = k
            break
        end
        append!(pos, (bits(world.player_position_wu[1]), bits(world.player_position_wu[2])))
        push!(dirs, world.player_direction_au); push!(rew, bits(Float32(world.reward))); push!(done, world.done ? 1 : 0)
        append!(goals, (world.goal_position[1], world.goal_position[2]))
    end
    open(joinpath(OUT, SEEDED_NAME * ".txt"), "w") do io
        println(io, "name ", SEEDED_NAME)
        println(io, "versions ", versions)
        println(io, "rng MersenneTwister(1) passed as SingleRoom(; rng)")
        println(io, "shape ", join_ints((H, W, N, world.num_directions, size(env.camera_view, 1), size(env.top_view, 1), size(env.top_view, 2))))
        println(io, "directions_wu_bits ", join_ints([bits(v[k]) for v in world.directions_wu for k in 1:2]))
        println(io, "reset_steps ", join_ints(SEEDED_RESET_STEPS))
        println(io, "reset_states ", join_ints(states))                          # (goal_i goal_j x_bits y_bits heading) x (1 + resets)
        println(io, "rollout_actions ", join_ints(actions))
        println(io, "rollout_error_step ", error_step)
        println(io, "rollout_position_bits ", join_ints(pos))
        println(io, "rollout_direction_au ", join_ints(dirs))
        println(io, "rollout_reward_bits ", join_ints(rew))
        println(io, "rollout_done ", join_ints(done))
        println(io, "rollout_goal ", join_ints(goals))
    end
    write(joinpath(OUT, SEEDED_NAME * ".camera_view_after_rollout.u32"), env.camera_view)
    println("wrote ", SEEDED_NAME)
end

function main()
    mkpath(OUT)
    cases = [split(chomp(l), '\t') for l in eachline(joinpath(HERE, "discriminator_cases.tsv")) if !startswith(l, "#")]
    versions = "julia=$(VERSION) RayCastWorlds=$(pkg_version(RCW)) RayCaster=$(pkg_version(SRM.RC)) " *
               "StaticArrays=$(pkg_version(SRM.SA)) SimpleDraw=$(pkg_version(SRM.SD))"
    open(joinpath(OUT, "manifest.tsv"), "w") do mf
        println(mf, "# ", versions)
        for c in cases
            name = String(c[1])
            H, W, N = parse(Int, c[2]), parse(Int, c[3]), parse(Int, c[4])
            gi, gj = parse(Int, c[5]), parse(Int, c[6])
            x, y = f32(c[7]), f32(c[8])
            d = parse(Int, c[9])
            pu = length(c) >= 10 ? parse(Int, c[10]) : 32                              # (optional column: pixels per tile of the top view)
            env = SRM.SingleRoom(height_tile_map_tu = H, width_tile_map_tu = W, num_rays = N, pu_per_tu = pu)
            inject!(env, gi, gj, x, y, d)
            world = env.world
            write(joinpath(OUT, name * ".camera_view.u32"), env.camera_view)
            write(joinpath(OUT, name * ".top_view.u32"), env.top_view)
            open(joinpath(OUT, name * ".txt"), "w") do io
                println(io, "name ", name)
                println(io, "versions ", versions)
                println(io, "shape ", join_ints((H, W, N, world.num_directions, size(env.camera_view, 1), size(env.top_view, 1), size(env.top_view, 2))))
                println(io, "directions_wu_bits ", join_ints([bits(v[k]) for v in world.directions_wu for k in 1:2]))
                println(io, "ray_direction_bits ", join_ints([bits(v[k]) for v in world.ray_directions_wu for k in 1:2]))
                println(io, "ray_stop_position_tu ", join_ints(vec(world.ray_stop_position_tu)))      # (2, N) column-major: i, j per ray
                println(io, "ray_hit_dimension ", join_ints(world.ray_hit_dimension))
                println(io, "ray_distance_bits ", join_ints(bits.(world.ray_distance_wu)))
                # a rollout from the injected state: act!(env, a) SR:333-340 under a fixed action stream
                actions = lcg_actions(64, UInt64(99))
                pos, dirs, rew, done = Int[], Int[], Int[], Int[]
                error_step = 0
                for (k, a) in enumerate(actions)
                    try
                        RCW.act!(env, a)
                    catch err
                        err isa BoundsError || rethrow()
                        error_step = k                      # the reachable BoundsError of collision_detection.jl:35
                        break
                    end
                    append!(pos, (bits(world.player_position_wu[1]), bits(world.player_position_wu[2])))
                    push!(dirs, world.player_direction_au)
                    push!(rew, bits(Float32(world.reward)))
                    push!(done, world.done ? 1 : 0)
                end
                println(io, "rollout_actions ", join_ints(actions))
                println(io, "rollout_error_step ", error_step)
                println(io, "rollout_position_bits ", join_ints(pos))
                println(io, "rollout_direction_au ", join_ints(dirs))
                println(io, "rollout_reward_bits ", join_ints(rew))
                println(io, "rollout_done ", join_ints(done))
            end
            write(joinpath(OUT, name * ".camera_view_after_rollout.u32"), env.camera_view)
            println(mf, name)
            println("wrote ", name)
        end
    end
    seeded_case(versions)
    println("done: ", length(cases), " cases + the seeded one in ", abspath(OUT))
end

main()
