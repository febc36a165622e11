/*
 * rcw.h — C ABI of librcw_hip: batched SingleRoom step/render on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of RayCastWorlds.jl: the SingleRoom
 * step/render hot path.  The reference has no FFI of its own on this path (it is
 * Julia calling Julia), so each entry point below cites the reference method it
 * replaces (paths relative to the reference tree; SR = src/single_room.jl).
 * The Julia-side binding a maintainer would add is shown in INTEGRATION.md and
 * shipped in julia/BatchedSingleRoom.jl.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types.
 *   - Every call returns RCW_OK (0) or a negative RCW_ERR_* code; the message for the
 *     last failure on the calling thread is available from rcw_last_error().
 *   - One caller thread per handle (the reference is single threaded).  All work of a
 *     handle is ordered on one HIP stream.  rcw_step, rcw_reset and rcw_set_state may
 *     return before the GPU has finished; every getter and rcw_sync() waits.
 *   - There is NO CPU fallback in this library: rcw_create fails with
 *     RCW_ERR_NO_DEVICE when no gfx950 device is usable.
 *   - Indices handed over the boundary are the reference's: tiles are 1-based (i, j),
 *     actions are 1..4, directions are 0..num_directions-1.
 *   - Batched layouts are the reference's single-agent (column-major) layouts with a
 *     trailing batch axis, i.e. what a Julia caller would unsafe_wrap:
 *       camera_view  UInt32 (H_cam, N, B)          SR:300
 *       tile_map     BitArray{3}(2, H, W).chunks   SR:54   -> UInt64 (nchunks, B)
 *       position     Float32 (2, B)                SR:24
 *       rays         Int64 (2, N, B), Int64 (N, B), Float32 (N, B), Float32 (2, N, B)
 *                                                   SR:29-31,39
 */
#ifndef RCW_H
#define RCW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCW_ABI_VERSION 4

#if defined(__GNUC__)
#define RCW_API __attribute__((visibility("default")))
#else
#define RCW_API
#endif

/* ---- error codes ------------------------------------------------------------------ */
#define RCW_OK                    0
#define RCW_ERR_INVALID_ARGUMENT -1  /* NULL pointer, bad size, bad config value          */
#define RCW_ERR_INVALID_ACTION   -2  /* action outside 1..4 (stands in for @assert SR:140) */
#define RCW_ERR_NO_DEVICE        -3  /* no usable gfx950 device / HIP runtime failure      */
#define RCW_ERR_OUT_OF_MEMORY    -4
#define RCW_ERR_OUT_OF_BOUNDS    -5  /* a tile index left the map (Julia: BoundsError at
                                        collision_detection.jl:35 or inside cast_ray)      */
#define RCW_ERR_HIP              -6  /* any other HIP error; text in rcw_last_error()      */
#define RCW_ERR_UNSUPPORTED      -7
/* A per-agent status WARNING (rcw_status), not an error: no call fails on it and the handle's error word stays clear.
 * sample_empty_position (utils.jl:23-37) gave up after max_tries = 1024*H*W occupied draws and reset!(world) placed the player
 * on the last — occupied — tile drawn; the reference @warns there (utils.jl:34) and goes on.  Only a map without an empty tile
 * (3x3: the one interior tile is the goal) gets here. */
#define RCW_WARN_SAMPLER_GAVE_UP  1

/* ---- objects, actions (SR:16-19) --------------------------------------------------- */
#define RCW_NUM_OBJECTS 2
#define RCW_WALL        1
#define RCW_GOAL        2
#define RCW_NUM_ACTIONS 4
#define RCW_ACTION_MOVE_FORWARD  1   /* SR:486 get_action_names */
#define RCW_ACTION_MOVE_BACKWARD 2
#define RCW_ACTION_TURN_LEFT     3
#define RCW_ACTION_TURN_RIGHT    4

/* Column colour ids used by the compact per-column descriptor (SR:417-429). */
#define RCW_COLOUR_WALL_DIM_1 0
#define RCW_COLOUR_WALL_DIM_2 1
#define RCW_COLOUR_GOAL_DIM_1 2
#define RCW_COLOUR_GOAL_DIM_2 3

/* The three choices inside the un-vendored dependencies that the reference's own tests
 * do not pin (DESIGN.md "parity unpinned").  They live here, in ONE
 * place, so a maintainer with a Julia toolchain can flip them without touching code;
 * tests/golden/discriminators.json holds poses where the two readings of each differ and
 * julia/make_reference_fixtures.jl dumps what the real package computes for them.
 * A fourth unpinned piece has no switch: the direction table (SR:65-69) is built with the C
 * library's cos / sin, Julia uses its own implementations — both < 1 ulp in Float64, so a
 * last-bit difference is possible (plausible for T = Float64, ~1e-8 per entry for Float32).
 * rcw_set_direction_table[64] lets the caller hand over Julia's table. */
#define RCW_DDA_TIE_X_FIRST_ON_LT 0  /* step in x when side_x <  side_y (default)          */
#define RCW_DDA_TIE_X_FIRST_ON_LE 1  /* step in x when side_x <= side_y                    */
#define RCW_DDA_DIST_SIDE_MINUS_DELTA 0 /* distance = side - delta after the step (default) */
#define RCW_DDA_DIST_PRE_INCREMENT    1 /* distance = side before it was incremented       */
#define RCW_NORMALIZE_INV_NORM_TIMES 0  /* inv(norm(v)) * v  (StaticArrays 1.2, default)    */
#define RCW_NORMALIZE_DIVIDE         1  /* v / norm(v)                                      */

/* What a move does when is_player_colliding (collision_detection.jl:30-35) would index a
 * tile off the map.  The reference raises BoundsError there — and this is reachable at the
 * defaults: walking along +x (heading 0) or +y (heading nd/4) in exact 1/8 steps reaches
 * x = H - 1 - 1/8, which does not collide (strict `<`, CD:18); the next forward move tests
 * x = H - 1, whose tile is H, and reads tile H + 1.
 *   RCW_OOB_ERROR        the faulting agent is left exactly as it was (as the Julia world is
 *                        after the exception), its status word and the handle's error word
 *                        are set to RCW_ERR_OUT_OF_BOUNDS (reported by the next
 *                        rcw_sync/getter, cleared by rcw_clear_error); other agents and
 *                        later steps are not affected.
 *   RCW_OOB_TREAT_EMPTY  tiles off the map count as empty; no error (the move above is then
 *                        simply blocked by the wall tile). */
#define RCW_OOB_ERROR       0
#define RCW_OOB_TREAT_EMPTY 1

/* R of SingleRoom(; R = ...) SR:266: the element type of world.reward / world.goal_reward (SR:33-34).  The
 * reference only ever stores zero(R) (SR:81, SR:131, SR:170-186) and goal_reward = one(R) (SR:82) in it. */
#define RCW_REWARD_FLOAT32 0   /* default */
#define RCW_REWARD_FLOAT64 1
#define RCW_REWARD_INT32   2
#define RCW_REWARD_INT64   3

/* Mirrors the keyword arguments of SingleRoom(; ...) SR:258-272 and the colour
 * constants SR:288-296.  Fill with rcw_config_default() and then override. */
typedef struct rcw_config {
    int32_t  abi_version;             /* RCW_ABI_VERSION                                   */
    int32_t  height_tile_map_tu;      /* SR:260  default 8  (x axis, index i)              */
    int32_t  width_tile_map_tu;       /* SR:261  default 16 (y axis, index j)              */
    int32_t  num_directions;          /* SR:262  default 128                               */
    int32_t  num_rays;                /* SR:268  default 512 (= camera view width)         */
    int32_t  height_camera_view_pu;   /* SR:271  default 256                               */
    int32_t  pu_per_tu;               /* SR:269  default 32 (pixels per tile of the top view) */
    float    player_radius_wu;        /* SR:263  default 1/8, must be in (0, 0.5)          */
    float    position_increment_wu;   /* SR:264  default 1/8                               */
    float    semi_field_of_view_wu;   /* SR:267  default Float32(2/3)                      */
    float    camera_height_tile_wu;   /* SR:270  default 1                                 */
    float    goal_reward;             /* SR:82   one(R) for R = Float32; other R: goal_reward_f64 */
    uint32_t floor_color;             /* SR:291  0x00404040                                */
    uint32_t ceiling_color;           /* SR:292  0x00FFFFFF                                */
    uint32_t wall_dim_1_color;        /* SR:293  0x00808080                                */
    uint32_t wall_dim_2_color;        /* SR:294  0x00c0c0c0                                */
    uint32_t goal_dim_1_color;        /* SR:295  0x00800000                                */
    uint32_t goal_dim_2_color;        /* SR:296  0x00c00000                                */
    int32_t  dda_tie_break;           /* RCW_DDA_TIE_*                                     */
    int32_t  dda_distance;            /* RCW_DDA_DIST_*                                    */
    int32_t  normalize_mode;          /* RCW_NORMALIZE_*                                   */
    int32_t  auto_reset;              /* 0 (reference behaviour: none) | 1: an agent whose
                                         `done` is set is re-sampled by the NEXT step call
                                         (its action is ignored, reward 0, done false)     */
    int64_t  agent_id_offset;         /* global id of local agent 0 (multi-GPU sharding);
                                         keys the reset RNG so results do not depend on
                                         how agents are sharded                            */
    int32_t  reward_type;             /* RCW_REWARD_*: R of SingleRoom(; R = ...) SR:266; the reward array
                                         (rcw_reward_typed, rcw_reward_device_ptr) has this element type  */
    int32_t  out_of_bounds;           /* RCW_OOB_ERROR (default, reference behaviour) |
                                         RCW_OOB_TREAT_EMPTY                               */
    int32_t  render_top_view;         /* 1: every reset/step also renders the top view
                                         (update_top_view! SR:446-483); default 0 — the
                                         reference always does, but it is a debug view, not the
                                         observation, and it doubles the bytes per step          */
    int32_t  world_unit_bits;         /* T of SingleRoom(; T = ...) SR:259: 32 (Float32, default) or 64
                                         (Float64).  With 64 the four world-unit parameters are
                                         taken from the *_f64 fields below (convert(T, 2/3) is
                                         not the Float32 value widened), positions / rays / tables
                                         cross the boundary as double (the *64 entry points), and
                                         every Float32 operation of the path becomes the same
                                         Float64 operation.  R (reward_type) is independent of T. */
    double   player_radius_wu_f64;    /* SR:263 convert(Float64, 1/8)                           */
    double   position_increment_wu_f64;
    double   semi_field_of_view_wu_f64;
    double   camera_height_tile_wu_f64;
    double   goal_reward_f64;         /* SR:82 one(R) for R = Float64 / Int32 / Int64 (converted to R)     */
    int32_t  reserved[2];
} rcw_config;

typedef struct rcw_handle rcw_handle;   /* opaque */

/* Reference defaults of SingleRoom(; ...) SR:258-272, SR:288-296. */
RCW_API int rcw_config_default(rcw_config* cfg);

/* SingleRoom(; kwargs...) SR:258-324 for `batch` independent agents on HIP device
 * `device` (>= 0).  Allocates structure-of-arrays state, the per-(direction, ray) table
 * (SR:65-69, SR:214-221) and the UInt32 (H_cam, N, B) observation batch in HBM.
 * The initial state is rcw_reset(h, NULL, seed). */
RCW_API int rcw_create(const rcw_config* cfg, int32_t batch, int32_t device, uint64_t seed,
               rcw_handle** out);
RCW_API int rcw_destroy(rcw_handle* h);

/* Replace the direction table directions_wu (SR:65-69) by a caller-computed one
 * (Float32 (2, num_directions)), e.g. Julia's own cos/sin, and rebuild the ray table. */
RCW_API int rcw_set_direction_table(rcw_handle* h, const float* directions_wu);

/* Use a caller-owned HIP stream (hipStream_t passed as void*); NULL restores the
 * handle's own stream.  The caller keeps the stream alive. */
RCW_API int rcw_set_stream(rcw_handle* h, void* hip_stream);
/* The stream the handle's work is ordered on (hipStream_t as void*), e.g. to make it wait for the
 * producer of device-resident actions or to order a consumer of the observations behind a step. */
RCW_API int rcw_get_stream(rcw_handle* h, void** hip_stream);
/* Render into a caller-owned DEVICE buffer of B*N*H_cam UInt32 instead of the
 * library's (NULL restores it).  Takes effect at the next render; does not synchronise, so a
 * caller can alternate two buffers while the previous frame batch is still being consumed. */
RCW_API int rcw_bind_obs(rcw_handle* h, void* device_ptr);

/* RCW.reset!(env) SR:326-331 -> SR:110-137 for the agents whose mask byte is non-zero
 * (mask == NULL: all).  Sampling is done on the device with a counter-based generator
 * keyed (seed, global agent id, episode number): goal uniform on interior tiles
 * (SR:120), player uniform on empty tiles by rejection (utils.jl:23-58) at the tile
 * centre (SR:125), heading uniform on 0..nd-1 (SR:128); reward 0, done false
 * (SR:131-132); rays cast and camera view rendered (SR:134, SR:329).
 * The stream is the build's own (Julia's RNG streams are not reproducible across
 * Julia versions): parity with the reference is in distribution only. */
RCW_API int rcw_reset(rcw_handle* h, const uint8_t* mask_host, uint64_t seed);

/* Inject the post-reset state the reference would have after SR:118-132 (this is how
 * "identical seeds" is realised): goal_ij Int32 (2, B) 1-based,
 * position_wu Float32 (2, B), direction_au Int32 (B); mask as above.  Clears the old
 * goal bit, sets the new one, zeroes reward/done, casts and renders. */
RCW_API int rcw_set_state(rcw_handle* h, const int32_t* goal_ij, const float* position_wu,
                  const int32_t* direction_au, const uint8_t* mask_host);

/* Float64 worlds (cfg.world_unit_bits = 64): the same calls with double arrays.  The Float32
 * forms return RCW_ERR_UNSUPPORTED on a Float64 handle and vice versa. */
RCW_API int rcw_set_state64(rcw_handle* h, const int32_t* goal_ij, const double* position_wu,
                            const int32_t* direction_au, const uint8_t* mask_host);
RCW_API int rcw_position64(rcw_handle* h, double* out_host /* (2, B) */);
RCW_API int rcw_rays64(rcw_handle* h, int32_t first, int32_t count, int64_t* stop_ij, int64_t* hit_dimension,
                       double* distance_wu, double* directions_wu);
RCW_API int rcw_set_direction_table64(rcw_handle* h, const double* directions_wu);
RCW_API int rcw_ray_table64(rcw_handle* h, double* out_host);
RCW_API int rcw_direction_table64(rcw_handle* h, double* out_host);

/* RCW.act!(env, action) SR:333-340 (minus update_top_view!) for every agent:
 * dynamics SR:139-191 -> cast_rays! SR:195-231 -> update_camera_view! SR:374-444.
 * actions: UInt8 (B), values 1..4, in HOST memory.  Any value outside 1..4 returns
 * RCW_ERR_INVALID_ACTION and NO agent is mutated (@assert SR:140). */
RCW_API int rcw_step(rcw_handle* h, const uint8_t* actions_host);
/* Same with actions already in DEVICE memory (stream-ordered, no host round trip).  Each
 * agent's action is checked on the device by the workgroup that steps it: an agent given a
 * value outside 1..4 is NOT stepped (its state and frame stay as they were), its status
 * word and the handle's error word are set to RCW_ERR_INVALID_ACTION, and the other agents
 * step normally — what a loop `for (env, a) in zip(envs, actions) act!(env, a)` over
 * reference worlds leaves behind, except that agents after the faulting one also run.
 * The error is returned by the next rcw_sync()/getter and cleared by rcw_clear_error. */
RCW_API int rcw_step_device(rcw_handle* h, const uint8_t* actions_device);

/* The three renderers of the reference as separate calls, for every agent, stream-ordered like rcw_step:
 *   rcw_cast_rays            RCW.cast_rays!(world) SR:195-231: recompute the rays (and with them the compact
 *                            column descriptors) from the current state; no pixel is written.
 *   rcw_update_camera_view   RCW.update_camera_view!(env) SR:374-444: refill camera_view from the stored ray
 *                            results, without casting (as in the reference, whose update_*_view! never cast).
 *   rcw_update_top_view      RCW.update_top_view!(env) SR:446-483 (needs cfg.render_top_view).
 * rcw_step / rcw_reset / rcw_set_state already do all of them; these exist because the reference exports them. */
RCW_API int rcw_cast_rays(rcw_handle* h);
RCW_API int rcw_update_camera_view(rcw_handle* h);
RCW_API int rcw_update_top_view(rcw_handle* h);

RCW_API int rcw_sync(rcw_handle* h);
RCW_API int rcw_clear_error(rcw_handle* h);

/* RLBase.state(env) SR:576: the camera view batch, aliased — the pointer is stable for
 * the handle's lifetime (or until rcw_bind_obs) and is overwritten by the next step. */
RCW_API int rcw_obs_device_ptr(rcw_handle* h, void** device_ptr);
/* Copy frames of agents [first, first+count) to host: UInt32 (H_cam, N, count). */
RCW_API int rcw_obs_copy(rcw_handle* h, uint32_t* out_host, int32_t first, int32_t count);
/* env.top_view SR:302 (needs cfg.render_top_view = 1): UInt32 (H*pu, W*pu, B), column-major,
 * as update_top_view! SR:446-483 leaves it: draw_tile_map! SR:342-372 (tile colours
 * 0x00FFFFFF wall / 0x00FF0000 goal / 0x00000000 free SR:288, grid 0x00cccccc), one line per
 * ray to its stop point SR:473-477 (0x00808080) and the player circle SR:480 (0x00c0c0c0).
 * The line and circle rasterisers belong to SimpleDraw 0.3 (un-vendored): Bresenham / midpoint
 * circle are ASSUMED — parity unpinned. */
RCW_API int rcw_top_view_device_ptr(rcw_handle* h, void** device_ptr);
RCW_API int rcw_top_view_copy(rcw_handle* h, uint32_t* out_host, int32_t first, int32_t count);
/* RLBase.reward SR:583 / RLBase.is_terminated SR:584.  rcw_reward serves R = Float32 handles; rcw_reward_typed
 * copies B elements of the handle's R (cfg.reward_type) whatever it is; the device array behind
 * rcw_reward_device_ptr has that element type too. */
RCW_API int rcw_reward(rcw_handle* h, float* out_host /* (B) */);
RCW_API int rcw_reward_typed(rcw_handle* h, void* out_host /* (B) of R */);
RCW_API int rcw_done(rcw_handle* h, uint8_t* out_host /* (B) */);
RCW_API int rcw_reward_device_ptr(rcw_handle* h, void** device_ptr);
RCW_API int rcw_done_device_ptr(rcw_handle* h, void** device_ptr);
/* world.player_position_wu SR:24, world.player_direction_au SR:25, world.goal_position SR:32 */
RCW_API int rcw_position(rcw_handle* h, float* out_host /* (2, B) */);
RCW_API int rcw_direction(rcw_handle* h, int32_t* out_host /* (B) */);
RCW_API int rcw_goal(rcw_handle* h, int32_t* out_host /* (2, B), 1-based */);
RCW_API int rcw_episode(rcw_handle* h, uint32_t* out_host /* (B): resets seen by each agent */);
/* Per-agent sticky status: 0, RCW_ERR_OUT_OF_BOUNDS (see RCW_OOB_ERROR),
 * RCW_ERR_INVALID_ACTION (rcw_step_device) or — a warning, where no error is recorded —
 * RCW_WARN_SAMPLER_GAVE_UP (utils.jl:34).  Does not fail on a set error word, so it can
 * be used to find the faulting agents. */
RCW_API int rcw_status(rcw_handle* h, int32_t* out_host /* (B) */);
/* world.tile_map SR:22 as BitArray{3}(2, H, W).chunks per agent: UInt64 (nchunks, B),
 * nchunks = rcw_tile_map_num_chunks(). Bit (o-1) + 2(i-1) + 2H(j-1), LSB first. */
RCW_API int rcw_tile_map_num_chunks(rcw_handle* h, int32_t* out);
RCW_API int rcw_tile_map_chunks(rcw_handle* h, uint64_t* out_host);
/* world.ray_stop_position_tu / ray_hit_dimension / ray_distance_wu / ray_directions_wu
 * SR:29-31,39 for agents [first, first+count), recomputed from the current state by a
 * cast-only kernel (the step itself does not spend HBM bandwidth on them).
 * Any output pointer may be NULL. */
RCW_API int rcw_rays(rcw_handle* h, int32_t first, int32_t count,
             int64_t* stop_ij /* (2, N, count) */, int64_t* hit_dimension /* (N, count) */,
             float* distance_wu /* (N, count) */, float* directions_wu /* (2, N, count) */);
/* Compact per-column descriptor of the current frames, indexed by image column k
 * (k = N - i + 1, SR:431): height_line_pu SR:408-411 (Int32, saturated) and colour id.
 * In the two-launch step these arrays are what the cast kernel hands to the fill kernel, refreshed by every step.  The one-launch
 * step (RCW_STEP_ONE_LAUNCH below) does not need them and every store of its casting workgroups costs the launch more than its bytes:
 * it refreshes them only for a caller that holds the device pointers — from the first rcw_columns_device_ptr call on, every step of
 * the handle does —; otherwise rcw_columns, the gathers and rcw_update_camera_view recast the current state (the cast kernel,
 * no action, ~10 us at 4096 agents x 256 columns) in front of their read, on demand. */
RCW_API int rcw_columns(rcw_handle* h, int32_t first, int32_t count,
                int32_t* height_line_pu /* (N, count) */, uint8_t* colour_id /* (N, count) */);
RCW_API int rcw_columns_device_ptr(rcw_handle* h, void** height_line_pu, void** colour_id);
/* Expand descriptors (DEVICE pointers, e.g. gathered from another GPU) into frames
 * UInt32 (H_cam, N, count) in DEVICE memory, with this handle's colours and H_cam:
 * the receiving side of the compact observation gather. */
RCW_API int rcw_expand_columns(rcw_handle* h, const int32_t* height_line_pu_device,
                       const uint8_t* colour_id_device, int32_t count, void* frames_device);

/* ---- multi-GPU: the observation gather north_star names, for a host without torch (Julia) ---------------
 * Agents shard by rank (cfg.agent_id_offset); stepping needs no communication.  The only exchange is the
 * optional all-gather of the observation batch over RCCL (xGMI inside a node).  librccl is loaded at run time
 * (dlopen of librccl.so.1 — the copy already in the process if there is one) the first time a comm call is
 * made, so single-GPU users do not need it.
 *   rcw_comm_unique_id   ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks
 *                        (a file, a socket, MPI, Julia's Distributed — the transport is the caller's).
 *   rcw_comm_init        ncclCommInitRank on the handle's device; collective over all `world` ranks.
 *   rcw_gather_columns   ncclAllGather of the compact descriptors (5 bytes per column) on the handle's stream:
 *                        height_line_pu Int32 (N, B*world) and colour id UInt8 (N, B*world) in DEVICE memory,
 *                        rank r's agents at [r*B, (r+1)*B).
 *   rcw_gather_observations   the GLOBAL observation batch UInt32 (H_cam, N, B*world) in caller-owned DEVICE
 *                        memory on every rank.  RCW_GATHER_COLUMNS: gather descriptors, expand to pixels locally
 *                        (rcw_expand_columns) — 5 bytes per column over the links instead of 4*H_cam;
 *                        RCW_GATHER_FRAMES: ncclAllGather of the pixels themselves.
 * All of them are stream-ordered behind the step that produced the frames and return without waiting. */
#define RCW_UNIQUE_ID_BYTES 128
#define RCW_GATHER_COLUMNS 0
#define RCW_GATHER_FRAMES  1
RCW_API int rcw_comm_unique_id(void* out_id /* RCW_UNIQUE_ID_BYTES */);
RCW_API int rcw_comm_init(rcw_handle* h, const void* unique_id, int32_t rank, int32_t world);
RCW_API int rcw_comm_destroy(rcw_handle* h);
RCW_API int rcw_comm_info(rcw_handle* h, int32_t* rank, int32_t* world /* 0, 0 before rcw_comm_init */);
RCW_API int rcw_gather_columns(rcw_handle* h, int32_t* height_line_pu_all_device, uint8_t* colour_id_all_device);
RCW_API int rcw_gather_observations(rcw_handle* h, int32_t mode, void* frames_all_device);

/* Device buffers for a host that has no GPU array package of its own (the gather outputs above, rcw_bind_obs,
 * rcw_expand_columns): plain hipMalloc / hipFree on the handle's device, and a copy to host memory that is
 * ordered behind the handle's stream (it waits for the work enqueued so far, then copies). */
RCW_API int rcw_device_malloc(rcw_handle* h, uint64_t bytes, void** device_ptr);
RCW_API int rcw_device_free(rcw_handle* h, void* device_ptr);
RCW_API int rcw_memcpy_to_host(rcw_handle* h, void* dst_host, const void* src_device, uint64_t bytes);

/* The (direction, ray) table the kernels use, for inspection and parity tests:
 * Float32 (N, 5, num_directions) = per direction [dx | dy | |1/dx| | |1/dy| | dir.ray]. */
RCW_API int rcw_ray_table(rcw_handle* h, float* out_host);
RCW_API int rcw_direction_table(rcw_handle* h, float* out_host /* (2, nd) */);

/* HIP-event timing on the handle's stream (what bench.py reads the kernel time from). */
RCW_API int rcw_timer_start(rcw_handle* h);
RCW_API int rcw_timer_stop(rcw_handle* h, float* elapsed_ms);

/* Per-kernel timing: while enabled, HIP events bracket the cast kernel, the top view kernel (when
 * cfg.render_top_view) and the fill kernel of each step (at most 256 steps are recorded).
 * rcw_profile_read returns their mean durations over the recorded steps (top_view_ms = 0 without it).
 * With the two-kernel top view (RCW_TOP_VIEW_TWO_KERNELS below) top_view_ms is its store kernel — the one that
 * writes the image; its drawing runs inside the camera fill's launch (or, at camera heights other than 256 rows, on a
 * side stream beside it): inside fill_ms.  Where the camera fill is the shorter of the two (big images) it is the FILL
 * that goes to the side stream: fill_ms then runs from the cast kernel's end to the fill's end beside the drawing, and
 * top_view_ms from there to the step's end; cast_ms + top_view_ms + fill_ms is the whole step in every case. */
RCW_API int rcw_profile(rcw_handle* h, int32_t enable);
RCW_API int rcw_profile_read(rcw_handle* h, float* cast_ms, float* top_view_ms, float* fill_ms, int32_t* steps);

/* Which form of update_top_view! (SR:446-483) the handle's geometry takes — all write the same pixels:
 *   RCW_TOP_VIEW_NONE         cfg.render_top_view = 0
 *   RCW_TOP_VIEW_IN_PLACE     tiles, then lines and circle drawn over them in HBM (images beyond the LDS bit planes)
 *   RCW_TOP_VIEW_ONE_KERNEL   write-once: bit planes in LDS, draw and store groups of one persistent kernel
 *   RCW_TOP_VIEW_TWO_KERNELS  write-once, inside rcw_step / rcw_reset / rcw_set_state: the drawing (bit planes -> HBM) in the
 *                             camera fill's own launch, then the moving-window store kernel; rcw_update_top_view alone
 *                             takes draw -> store back to back where that is the faster one (rcw_update_top_view_form).
 *                             Geometries: H*pu a multiple of 256 rows with pu_per_tu in {8, 16, ..., 256} and a player
 *                             circle of <= 32 rows (rcw_top_store_kernel), any pu_per_tu >= 9 with H*pu a multiple of 4
 *                             and >= 42 rows (rcw_top_store_flat_kernel), 8-pixel tiles with H*pu a multiple of 64 or 32
 *                             (rcw_top_store_units_kernel).  Taken at every batch size where the camera view is 256
 *                             rows high (one launch for fill + drawing); at other camera heights the drawing needs the
 *                             handle's side stream, and the form is taken from 256 MiB of top view a step (below that
 *                             the one-kernel form is faster) unless rcw_set_top_view_form asks for it */
enum { RCW_TOP_VIEW_NONE = 0, RCW_TOP_VIEW_IN_PLACE = 1, RCW_TOP_VIEW_ONE_KERNEL = 2, RCW_TOP_VIEW_TWO_KERNELS = 3 };
RCW_API int rcw_top_view_form(rcw_handle* h, int32_t* form);
/* ... and the form rcw_update_top_view takes when it is called ALONE (update_top_view!(env) outside a step, SR:446): nothing runs
 * beside the drawing then.  Where the step takes the two-kernel form the stand-alone call takes it too (draw -> store back to back)
 * for images from 256 x 256 px, for pixel scales that are no multiple of 4 and for tiles below 16 px — measured faster, round 5 —;
 * the one-kernel form keeps images below 256 x 256 px at 16, 20, 24, ... px a tile, and every geometry that is not the two-kernel
 * form's; RCW_TOP_VIEW_IN_PLACE where the bit planes do not fit in LDS. */
RCW_API int rcw_update_top_view_form(rcw_handle* h, int32_t* form);
/* Choose the form instead of the rule above (all forms write the same pixels; this is a performance choice, e.g. the
 * one-kernel form for a caller that does not want the handle's side stream, or the two-kernel form for a small batch):
 * form = 0 restores the automatic choice, else RCW_TOP_VIEW_IN_PLACE / ONE_KERNEL / TWO_KERNELS; RCW_ERR_UNSUPPORTED when
 * the geometry cannot take it (the handle then keeps the automatic choice).  runs = 0: automatic; 1..8: the two-kernel
 * form draws and stores the batch in that many runs of agents.  Waits for the handle's stream; reallocates the form's
 * scratch in HBM.  The library reads no environment variable other than RCW_RCCL_LIBRARY: what used to be development
 * switches (RCW_TOP_SPLIT, RCW_TOP_RUNS, ...) exists only in the development build (make dev -> librcw_hip_dev.so). */
RCW_API int rcw_set_top_view_form(rcw_handle* h, int32_t form, int32_t runs);
/* How many launches a step — RCW.act!(env, a) SR:333-340 — takes; both forms leave the same state and the same pixels:
 *   RCW_STEP_TWO_LAUNCHES  the cast kernel (dynamics SR:139-191, cast_rays! SR:195-231, the columns of update_camera_view!
 *                          SR:401-429 as compact descriptors), then the fill kernel (SR:431-440) — the reference's own order.
 *   RCW_STEP_ONE_LAUNCH    the frame of the NEXT step depends only on this step's state and the next action, and there are four
 *                          actions: the casting workgroups of a launch commit the dynamics and also cast the four successor states
 *                          (a blocked / goal / raising move keeps the current frame; an agent that is done under cfg.auto_reset gets
 *                          the re-sampled world's, drawn ahead without being committed) into slots in HBM (two buffers of 10 bytes a
 *                          view column), and the fill workgroups of the NEXT launch, in the same launch as that step's casting, write the
 *                          frames the actions select.  Nothing inside a launch waits for anything else in it; the cast kernel and a
 *                          launch boundary leave the step's critical path.  rcw_reset / rcw_set_state (and a first step) cast as a
 *                          launch of their own, which leaves the slots of the agents it touches ready.
 * The rule: one launch where the geometry allows — a camera view the moving-window fill kernels take (256 rows: every BASELINE configuration;
 * also 256 k, 128 and 64 rows, up to 8191) without a top view (cfg.render_top_view = 0), fewer than 2^29 view columns in the batch — AND the batch is large enough for it to pay: a casting
 * workgroup marches five fans one after the other, so the launch is no shorter than that, and below ~100-350 MiB of frames a step
 * (by map size and view columns: 512 agents at 8x8 / 256 columns, 400 at 32x32 / 1024 columns) the cast kernel followed by the fill
 * kernel is the faster step (rcw_set_step_form(RCW_STEP_ONE_LAUNCH) takes it at any batch).  A step captured into a HIP graph (hipStreamBeginCapture on
 * the handle's stream) takes the two-launch form, and so does every later step of that handle: the one-launch form alternates its
 * two buffers from launch to launch on the host, which a replayed graph cannot.  rcw_set_step_form: 0 = the rule,
 * RCW_STEP_ONE_LAUNCH fails with RCW_ERR_UNSUPPORTED where the geometry cannot take it. */
enum { RCW_STEP_TWO_LAUNCHES = 1, RCW_STEP_ONE_LAUNCH = 2 };
RCW_API int rcw_step_form(rcw_handle* h, int32_t* form);
RCW_API int rcw_set_step_form(rcw_handle* h, int32_t form);
/* The kernel update_camera_view! (SR:374-444) runs in INSIDE A STEP of this handle (what bench.py labels its roofline block
 * with): "rcw_fill256_kernel", "rcw_fill_window_kernel", "rcw_fill_flat_kernel", ... by camera height and batch — and
 * "rcw_fill256_draw_kernel" where the handle also renders the top view and the camera fill and the top view's drawing go in
 * one launch, and "rcw_fill256_cast_kernel" for the one-launch step (rcw_update_camera_view alone always takes the plain fill
 * kernel). */
RCW_API int rcw_fill_kernel_name(rcw_handle* h, char* buf, int32_t buflen);

/* Introspection */
RCW_API int rcw_batch(rcw_handle* h, int32_t* out);
RCW_API int rcw_get_config(rcw_handle* h, rcw_config* out);
RCW_API int rcw_device_name(rcw_handle* h, char* buf, int32_t buflen);
RCW_API const char* rcw_last_error(void);
RCW_API int rcw_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RCW_H */
