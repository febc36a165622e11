// Device-side view of one batch of SingleRoom agents and the kernel launchers.
// State is structure-of-arrays in HBM, resident for the handle's lifetime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Per-direction ray table: for heading d the slice is 5 rows of N floats
//   [dx | dy | |1/dx| | |1/dy| | dir . ray]
// so lane i of the cast phase reads five coalesced floats (SR:214-221, SR:404).
#define RCW_TABLE_ROWS 5

struct RcwDev {
    // geometry / config (all wave-uniform, live in SGPRs)
    int32_t B, H, W, N, nd, Hc;
    int32_t nwords;          // 32-bit words of one agent's tile_map (2 per UInt64 chunk)
    int32_t real64;          // world-unit type T: 0 = Float32, 1 = Float64 (SR:259); the *64 members below
    float radius, radius_sq; // player_radius_wu, fl(r*r)            (CD:18)
    float inc;               // position_increment_wu                (UT:16-17)
    float goal_reward;       // SR:82 one(R), R = Float32
    double goal_reward64;    // one(R) for the other reward types (converted on store)
    int32_t reward_type;     // RCW_REWARD_*: element type R of `reward` (SR:33, SR:266)
    float num;               // fl(camera_height_tile_wu * N)        (SR:406)
    float two_fov;           // fl(2 * semi_field_of_view_wu)        (SR:406)
    double radius64, radius_sq64, inc64, num64, two_fov64;   // the same five in Float64
    uint32_t floor_color, ceiling_color;
    uint32_t colour[4];      // indexed by RCW_COLOUR_*              (SR:293-296)
    int32_t tie_le;          // RCW_DDA_TIE_X_FIRST_ON_LE
    int32_t dist_pre;        // RCW_DDA_DIST_PRE_INCREMENT
    int32_t auto_reset;
    int32_t oob_empty;       // RCW_OOB_TREAT_EMPTY
    int32_t fill_grid;       // workgroups of the fill kernel (the moving window = fill_grid KiB x 4)
    int32_t fill_plain;      // 1: plain stores, 0: non-temporal
    int32_t fill_flat;       // development only (RCW_FILL_FLAT): rcw_fill_flat_kernel also where rcw_fill_window_kernel<2 / 4> applies
    int32_t cast_block;      // threads per agent in the cast kernel (multiple of 64, <= 256)
    int32_t cast_ballot;     // development only (RCW_CAST_MARCH=ballot): the ballot-bounded march instead of the exec-masked one
    int32_t cast_table_lds;  // development only (RCW_CAST_TABLE=lds): stage the heading's ray-table slice in LDS first
    int32_t cast_waves;      // development only (RCW_CAST_WAVES=1): the cast kernel with a wavefront per agent, four agents a workgroup — measured, rejected
    int32_t cast_r3;         // development only (RCW_CAST_KERNEL=r3): the round-3 cast kernel (five dependent round trips), for the comparison
    int64_t agent_id_offset;
    uint64_t seed;
    // state (SR:21-40), one entry per agent
    float2* pos;             // player_position_wu (T = Float32)
    double2* pos64;          // player_position_wu (T = Float64)
    int32_t* dir;            // player_direction_au
    int2* goal;              // goal_position (1-based i, j)
    void* reward;            // R[B]
    uint8_t* done;
    uint32_t* episode;       // resets seen (keys the generator)
    uint32_t* tile_map;      // BitArray{3}(2,H,W).chunks viewed as 32-bit words, [B][nwords]
    // constants
    const float2* dir_table; // directions_wu [nd]
    const float* ray_table;  // [nd][RCW_TABLE_ROWS][N]
    const double2* dir_table64;   // the two tables in Float64
    const double* ray_table64;
    // outputs
    uint32_t* obs;           // camera_view UInt32 (Hc, N, B)
    int32_t* col_h;          // (N, B) height_line_pu by image column
    uint8_t* col_c;          // (N, B) colour id by image column
    uint32_t* top_view;      // optional env.top_view UInt32 (H*pu, W*pu, B)  SR:302
    int32_t pu;              // pu_per_tu
    int32_t top_rp;          // player_radius_pu = wu_to_pu(player_radius_wu, pu)  SR:469 (host-computed in T)
    int32_t top_lds;         // write-once LDS bit-plane kernel: the number of buffers in its ring (1..3); 0: in-place fallback
    int32_t top_grid;        // workgroups of the (persistent) write-once top view kernel
    int32_t top_unit_px;     // ... its store kernel's unit: 256 rows of an image column (a whole 1 KiB chunk), 128 or 64
    int32_t top_split;       // 1: the two-kernel top view (draw kernel -> planes in HBM -> moving-window store kernel)
    int32_t top_flat;        // ... with rcw_top_store_flat_kernel (any pixel scale >= 9): the image columns a 256-pixel chunk may touch; 0: the unit kernels
    int32_t top_plane_words; // ... and the words of one agent's region of top_plane in that form
    int32_t top_alone_split; // rcw_update_top_view alone (no camera fill beside it) also takes the two-kernel form, back to back
    int32_t top_runs;        // the batch is drawn and stored in this many runs of agents (store of run r beside the drawing of run r + 1)
    int32_t top_parts;       // draw workgroups an agent (two-kernel form on the side-stream / stand-alone path with rcw_top_store_kernel): 1, or 2..4 for few big images
    int32_t top_draw_first;  // (host) inside a step the drawing stays on the handle's stream and the camera fill goes to the side stream
    int32_t top_fused;       // a step's camera fill and top-view drawing go in ONE launch (rcw_fill256_draw_kernel) instead of two streams
    int32_t top_draw_block;  // threads of a draw-kernel workgroup: 256; a lane per ray (up to 1024) when the plane leaves room for few workgroups on a CU
    int32_t top_draw_block_alone;   // ... in rcw_update_top_view alone: 64 / 128 for batches of tens of thousands of small images
    int32_t top_store_plain; // its store kernel: 1 plain stores, 0 non-temporal
    int32_t top_store_grid;  // ... and its workgroups (the moving window = top_store_grid KiB x 4)
    uint32_t* top_plane;     // [B][W*pu][H*pu/32] ray-line bit plane of every agent (two-kernel top view)
    int2* top_hdr;           // [B] the player's pixel (ip, jp), 1-based  SR:468
    uint2* top_codes;        // [B][W][H*pu/256] 2-bit fill codes of a chunk's tiles
    // the store kernel FOLLOWS the draw kernel (round 5): both run at once on two streams, ordered through memory instead of an event
    uint32_t* top_flags;     // [blocks of 2^top_blk_shift agents] how many of the block's agents' planes, headers and codes were written, over all calls
    int32_t top_blk_shift;
    uint32_t top_epoch;      // this call's number among the calls that count: a complete block stands at top_epoch x its agents
    int32_t top_signal;      // draw kernel: publish each agent as it is done
    int32_t top_follow;      // store kernels: wait for the agents of a group of chunks before loading anything of theirs
    int32_t top_follow_ok;   // (host) the geometry's draw and store workgroups fit on a CU together, in a step (bit 0) / alone (bit 1)
    int32_t top_rotate;      // rcw_top_store_flat_kernel's wavefront -> chunk assignment turns by this many slots from group to group (33; development: RCW_TOP_ROTATE)
    int32_t fill_trips;      // development only (RCW_FILL_TRIPS=0..3): rcw_fill256_kernel's body with that many more dependent round trips a prefetch; -1: the kernel proper
    int32_t step_fused;      // development only (RCW_STEP_FUSED=1): cast and camera fill in ONE launch (rcw_step256_kernel), handed off through the two arrays below
    uint32_t* step_flags;    // ... [B] the epoch of the last step whose descriptors of this agent are complete
    uint32_t* step_hc;       // ... (N, B) the column's padding (SR:436, 0..256) | colour id << 9 | the step's epoch << 11 by image column
    uint32_t step_epoch;     // ... this launch's epoch (the handle counts its fused steps)
    int32_t top_debug;       // development only (RCW_TOP_DEBUG): bit 0 skip drawing, bit 1 skip storing — for timing the halves
    int32_t top_draw_banks;  // development only (RCW_TOP_DRAW=banks): wavefronts of one kind of line (axis, direction), every lane starting on its own LDS bank
    int32_t top_draw_r4;     // development only (RCW_TOP_DRAW=r4): the round-4 body of the draw kernel, for the comparison
    int32_t fill_pairs;      // development only (RCW_FILL_FLAT_PAIRS=1): rcw_fill_flat_kernel with two wavefronts to a slot of the window
    int32_t* err;            // sticky error word of the handle (0 = ok); never blocks a step
    int32_t* status;         // per-agent sticky status
#ifdef RCW_DEV_SWITCHES
    int32_t spec_debug;      // development only (RCW_SPEC_DEBUG): timing probes of the one-launch step — bit 0 the casting workgroups return at once, bit 1 the fill's (wrong frames)
#endif
};

struct RcwRayOut {           // rcw_rays(): SR:29-31,39 for agents [first, first+count)
    int64_t* stop_ij;        // (2, N, count)
    int64_t* hit_dim;        // (N, count)
    void* dist;              // (N, count) in T
    void* dirs;              // (2, N, count) in T
};

size_t rcw_step_lds_bytes(const RcwDev& p);

// act!(env, a) = cast kernel (dynamics + rays + projection -> column descriptors) followed by
// the fill kernel (descriptors -> pixels).  actions == nullptr: render only (after reset /
// set_state); mask == nullptr: all agents.
hipError_t rcw_launch_cast(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev,
                           hipStream_t s, int first = 0, int count = -1);   // agents [first, first + count); -1: to the end
hipError_t rcw_launch_fill(const RcwDev& p, const int32_t* col_h, const uint8_t* col_c, uint32_t* frames,
                           long long total_cols, const uint8_t* mask_dev, hipStream_t s);
// update_top_view!(env) SR:446-483 for every (unmasked) agent; needs p.top_view
hipError_t rcw_launch_top_view(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s);
// LDS bytes of the write-once top view kernel for this geometry, and the one-off preparation (raises the
// kernel's dynamic LDS limit when the bit planes need more than 64 KiB); sets nothing on the device.
size_t rcw_top_view_lds_bytes(const RcwDev& p);
// the two-kernel top view: eligibility of a geometry, its HBM scratch sizes, and the two launches
int rcw_top_split_unit(const RcwDev& p);   // rows of a store-kernel unit (256 / 128 / 64), 0: geometry not taken
int rcw_top_flat_cols(const RcwDev& p);    // rcw_top_store_flat_kernel: columns a chunk may touch, 0: geometry not taken
int32_t rcw_top_plane_words(const RcwDev& p);
int rcw_fill_flat_cols(const RcwDev& p);   // rcw_fill_flat_kernel: columns a chunk may touch at this camera height, 0: not taken
const char* rcw_fill_kernel_name(const RcwDev& p, long long total_cols);   // the kernel rcw_launch_fill takes
int rcw_fill_takes_256(const RcwDev& p, long long total_cols);              // ... is rcw_fill256_kernel (what the fused launches build on)
int rcw_fill_window_columns(const RcwDev& p, long long total_cols);         // ... 0: rcw_fill256_kernel, 1 / 2 / 4: rcw_fill_window_kernel<M>, -1: another one (the one-launch step takes the first four)
size_t rcw_top_plane_bytes(const RcwDev& p);
size_t rcw_top_codes_bytes(const RcwDev& p);
struct RcwHw { int cus, lds_per_cu, waves_per_cu; };       // what the top view's rule needs of the device (hipDeviceProp_t: multiProcessorCount, sharedMemPerBlock, maxThreadsPerMultiProcessor / 64)
int rcw_top_draw_per_cu(const RcwDev& p, int draw_block, int lds_per_cu = 160 * 1024, int waves = 28);   // draw workgroups resident on a CU together: by LDS, by the wavefront slots the camera fill leaves
int rcw_top_follow_fits(const RcwDev& p, int draw_block, bool beside_fill, int cus);   // draw + store (+ camera fill) workgroups resident on one CU together
hipError_t rcw_launch_top_draw(const RcwDev& p, const uint8_t* mask_dev, int first, int count, hipStream_t s, int block = 0);    // agents [first, first + count); block: threads a workgroup, 0 = p.top_draw_block
hipError_t rcw_launch_top_store(const RcwDev& p, const uint8_t* mask_dev, int first, int count, hipStream_t s);
// the one-launch step (round 6): eligibility of a geometry, the bytes of one of its two slot buffers ([B][5][N] packed column words), the launch
int rcw_step_spec_eligible(const RcwDev& p);
size_t rcw_step_spec_slot_bytes(const RcwDev& p);
hipError_t rcw_launch_step_spec(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev, const uint16_t* slots_in,
                                uint16_t* slots_out, bool with_fill, bool cols, hipStream_t s);   // cols: the step also leaves the current frame's (height, colour id) descriptors
#ifdef RCW_DEV_SWITCHES
bool rcw_step_fusable(const RcwDev& p);       // development experiment (RCW_STEP_FUSED): cast + camera fill in one launch
hipError_t rcw_launch_step256(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev, uint32_t epoch, hipStream_t s);
#endif
int rcw_fill_draw_fusable(const RcwDev& p);   // a step's camera fill + top-view drawing in one launch: this geometry takes it
hipError_t rcw_launch_fill256_draw(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s);   // (fills p.obs from p.col_h / p.col_c, draws every agent)
hipError_t rcw_prepare_top_view(const RcwDev& p, int device);
hipError_t rcw_launch_reset(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s);
hipError_t rcw_launch_set_state(const RcwDev& p, const int2* goal, const void* pos /* float2* or double2* */,
                                const int32_t* dir, const uint8_t* mask_dev, hipStream_t s);
hipError_t rcw_launch_init_tile_map(const RcwDev& p, hipStream_t s);
hipError_t rcw_launch_rays(const RcwDev& p, int32_t first, int32_t count, RcwRayOut out,
                           hipStream_t s);
hipError_t rcw_launch_expand(const RcwDev& p, const int32_t* col_h, const uint8_t* col_c,
                             int32_t count, uint32_t* frames, hipStream_t s);
