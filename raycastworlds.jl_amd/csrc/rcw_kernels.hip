// HIP kernels (gfx950 / CDNA4) for the batched SingleRoom step/render path.
//
// One workgroup (256 threads = 4 wavefronts) owns one agent for one step:
//   phase 0  the agent's tile_map (2·H·W bits) is staged in LDS; lane 0 runs the dynamics
//            (act!(world, a) SR:139-191, or the opt-in re-sample SR:110-137) and publishes
//            the new pose through LDS;
//   phase 1  one lane per view column: table lookup of the ray (SR:214-221), grid DDA
//            against the LDS tile map (RayCaster.cast_ray, SR:223), perpendicular distance
//            and column height (SR:404-411), colour (SR:417-429) -> an 8-byte column
//            descriptor in LDS, mirrored to image column k = N - i + 1 (SR:431);
//   phase 2  the workgroup re-maps lanes along the image's contiguous axis (rows of one
//            column, Julia column-major (H_cam, N)) and streams the frame out with
//            16-byte non-temporal stores — one wavefront store instruction writes one
//            whole 1 KiB column at H_cam = 256.
// This is an integer/indexing + streaming-store path: no MFMA, the roofline is HBM write
// bandwidth, and the frame (4·H_cam·N bytes per agent-step) is written exactly once.
//
// Floating point: every operation below is a single IEEE-754 Float32 rounding, exactly
// as the reference (Julia never contracts a*b+c): this file MUST be compiled with
// -ffp-contract=off and without fast-math; division and sqrt are the correctly rounded
// forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), denormals are kept.
#include "rcw_kernels.h"
#include "rcw_rng.h"
#include "../../include/rcw.h"

#include <limits.h>

namespace {

constexpr int kBlock = 256;   // 4 wavefronts of 64

// 16-byte store unit (a native vector, so __builtin_nontemporal_store accepts it)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- tile map (BitArray{3}(2,H,W): bit (o-1) + 2(i-1) + 2H(j-1))  SR:54 -------------
// 2 bits of tile (i, j), 1-based: bit 0 = WALL layer, bit 1 = GOAL layer.
__device__ __forceinline__ uint32_t tile_bits(const uint32_t* tm, int H, int i, int j)
{
    const int t = (i - 1) + H * (j - 1);
    return (tm[t >> 4] >> ((t & 15) * 2)) & 3u;
}
__device__ __forceinline__ void set_goal_bit(uint32_t* tm, int H, int i, int j, bool v)
{
    const int t = (i - 1) + H * (j - 1);
    const uint32_t m = 2u << ((t & 15) * 2);
    if (v) tm[t >> 4] |= m; else tm[t >> 4] &= ~m;
}

// ---- is_player_colliding for both layers in one sweep  (CD:21-42) ----------------------
// Per layer the reference walks the 3x3 tiles (j outer, i inner), indexes the map first
// (BoundsError if the tile is off the map) and returns at the first hit.  Result per
// layer: 0 = false, 1 = true, 2 = BoundsError.
struct Collide { int wall, goal; };
__device__ __forceinline__ Collide player_colliding(const uint32_t* tm, int H, int W, float px,
                                                    float py, float radius_sq)
{
    const int it = (int)floorf(px) + 1;   // wu_to_tu UT:5
    const int jt = (int)floorf(py) + 1;
    int wall = -1, goal = -1;             // -1 = undecided
    for (int j = jt - 1; j <= jt + 1; ++j) {
        for (int i = it - 1; i <= it + 1; ++i) {
            if (i < 1 || i > H || j < 1 || j > W) {
                if (wall < 0) wall = 2;
                if (goal < 0) goal = 2;
                continue;
            }
            const uint32_t bits = tile_bits(tm, H, i, j);
            if (bits == 0u) continue;
            const float cx = (float)i - 0.5f, cy = (float)j - 0.5f;      // CD:33-34
            const float qx = px - cx, qy = py - cy;                      // CD:35
            const float sx = qx < -0.5f ? -0.5f : (qx > 0.5f ? 0.5f : qx);   // CD:11
            const float sy = qy < -0.5f ? -0.5f : (qy > 0.5f ? 0.5f : qy);
            const float vx = qx - sx, vy = qy - sy;                      // CD:16
            const float vx2 = vx * vx, vy2 = vy * vy;
            const bool hit = (vx2 + vy2) < radius_sq;                    // CD:18
            if (hit) {
                if ((bits & 1u) && wall < 0) wall = 1;
                if ((bits & 2u) && goal < 0) goal = 1;
            }
        }
    }
    Collide c;
    c.wall = wall < 0 ? 0 : wall;
    c.goal = goal < 0 ? 0 : goal;
    return c;
}

// ---- reset!(world)  SR:110-137 with the counter-based generator -------------------------
// tm_a / tm_b: the agent's tile map words in up to two places (LDS copy and HBM).
struct Pose { float x, y; int d; };
__device__ __forceinline__ Pose reset_agent(const RcwDev& p, int a, uint32_t* tm_a, uint32_t* tm_b)
{
    const int H = p.H, W = p.W;
    const uint32_t ep = p.episode[a];
    const uint64_t key = rcw_episode_key(p.seed, (uint64_t)(p.agent_id_offset + a), (uint64_t)ep);
    uint64_t n = 0;
    const int2 old = p.goal[a];
    set_goal_bit(tm_a, H, old.x, old.y, false);                               // SR:118
    if (tm_b) set_goal_bit(tm_b, H, old.x, old.y, false);
    const int gi = 2 + (int)rcw_below(rcw_draw(key, n++), (uint64_t)(H - 2));  // SR:120
    const int gj = 2 + (int)rcw_below(rcw_draw(key, n++), (uint64_t)(W - 2));
    p.goal[a] = make_int2(gi, gj);                                            // SR:121
    set_goal_bit(tm_a, H, gi, gj, true);                                      // SR:122
    if (tm_b) set_goal_bit(tm_b, H, gi, gj, true);
    // sample_empty_position UT:52-58 -> UT:23-37: rejection over all H*W tiles
    const uint64_t HW = (uint64_t)H * (uint64_t)W;
    const uint64_t max_tries = 1024ull * HW;
    uint64_t lin = rcw_below(rcw_draw(key, n++), HW);                          // UT:24
    for (uint64_t t = 0; t < max_tries; ++t) {                                 // UT:26
        const int ti = (int)(lin % (uint64_t)H) + 1, tj = (int)(lin / (uint64_t)H) + 1;
        if (tile_bits(tm_a, H, ti, tj)) lin = rcw_below(rcw_draw(key, n++), HW);   // UT:27-28
        else break;
    }
    const int pi = (int)(lin % (uint64_t)H) + 1, pj = (int)(lin / (uint64_t)H) + 1;
    Pose o;
    o.x = (float)((double)pi - 0.5);                                          // SR:125
    o.y = (float)((double)pj - 0.5);
    o.d = (int)rcw_below(rcw_draw(key, n++), (uint64_t)p.nd);                  // SR:128
    p.pos[a] = make_float2(o.x, o.y);                                         // SR:126
    p.dir[a] = o.d;                                                           // SR:129
    p.reward[a] = 0.0f;                                                       // SR:131
    p.done[a] = 0;                                                            // SR:132
    p.episode[a] = ep + 1;
    return o;
}

// ---- RayCaster.cast_ray  (external; call site SR:223).  UNPINNED choices via p.tie_le /
// p.dist_pre (include/rcw.h).  Leaves the map -> oob (Julia: BoundsError). ------------------
struct RayHit { int i, j, dim; float dist; uint32_t bits; bool oob; };
__device__ __forceinline__ RayHit cast_ray(const uint32_t* tm, int H, int W, float x, float y,
                                           float dx, float dy, float ddx, float ddy, int tie_le,
                                           int dist_pre)
{
    int i = (int)floorf(x) + 1;
    int j = (int)floorf(y) + 1;
    int si, sj;
    float sx, sy;
    if (dx < 0.0f) { si = -1; sx = (x - (float)(i - 1)) * ddx; }
    else           { si = +1; sx = ((float)i - x) * ddx; }
    if (dy < 0.0f) { sj = -1; sy = (y - (float)(j - 1)) * ddy; }
    else           { sj = +1; sy = ((float)j - y) * ddy; }
    RayHit r;
    r.dim = 0; r.dist = 0.0f; r.bits = 0u; r.oob = false;
    // Every iteration moves one tile in a fixed direction, so the loop leaves the map (and
    // exits) after at most H + W steps even on a map without a closed wall ring.
    for (;;) {
        if ((unsigned)(i - 1) >= (unsigned)H || (unsigned)(j - 1) >= (unsigned)W) { r.oob = true; break; }
        r.bits = tile_bits(tm, H, i, j);
        if (r.bits) break;
        const bool x_first = tie_le ? (sx <= sy) : (sx < sy);
        if (x_first) { r.dist = sx; sx = sx + ddx; i += si; r.dim = 1; }
        else         { r.dist = sy; sy = sy + ddy; j += sj; r.dim = 2; }
    }
    if (!dist_pre) {
        if (r.dim == 1) r.dist = sx - ddx;
        else if (r.dim == 2) r.dist = sy - ddy;
    }
    r.i = i; r.j = j;
    return r;
}

// ---- column height  SR:404-411 ------------------------------------------------------------
__device__ __forceinline__ int height_line_pu(const RcwDev& p, float dist, float dot)
{
    const float projected = dist * dot;               // SR:404
    const float den = p.two_fov * projected;          // (2 * fov) * projected
    const float height_line = p.num / den;            // SR:406
    if (!isfinite(height_line)) return p.Hc;          // SR:410
    const float f = floorf(height_line);              // floor(Int, .) SR:408, saturated
    if (f >= 2147483648.0f) return INT_MAX;
    if (f <= -2147483648.0f) return INT_MIN;
    return (int)f;
}
__device__ __forceinline__ int column_padding(int Hc, int h)
{
    if (h >= Hc - 1) return 0;                        // SR:433 whole column = colour
    const long long pad = ((long long)Hc - (long long)h) / 2;   // SR:436
    return pad > (long long)Hc ? Hc : (int)pad;
}
// pixel of 0-based row r: rows [0,pad) ceiling, [pad,Hc-pad) colour, rest floor  SR:437-439
__device__ __forceinline__ uint32_t pixel(int r, int pad, int Hc, uint32_t colour, uint32_t ceil_c,
                                          uint32_t floor_c)
{
    return r < pad ? ceil_c : (r < Hc - pad ? colour : floor_c);
}

// ---- the step kernel --------------------------------------------------------------------
template <bool HC256>
__global__ __launch_bounds__(kBlock) void rcw_step_kernel(const RcwDev p,
                                                          const uint8_t* __restrict__ actions,
                                                          const uint8_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int a = blockIdx.x;
    const int tid = threadIdx.x;
    if (*p.err != 0) return;                   // sticky error: nothing is mutated
    if (mask != nullptr && mask[a] == 0) return;

    const int nwords_pad = (p.nwords + 3) & ~3;
    uint32_t* tm = lds;                                     // [nwords]
    int* s_pad = (int*)(lds + nwords_pad);                  // [N]
    uint32_t* s_col = lds + nwords_pad + p.N;               // [N]
    __shared__ float s_pose[4];

    uint32_t* tm_hbm = p.tile_map + (size_t)a * p.nwords;
    for (int w = tid; w < p.nwords; w += kBlock) tm[w] = tm_hbm[w];
    __syncthreads();

    // ---- phase 0: dynamics (lane 0) -------------------------------------------------------
    if (tid == 0) {
        const int act = actions ? (int)actions[a] : 0;
        float2 pos = p.pos[a];
        int d = p.dir[a];
        if (act != 0 && p.auto_reset && p.done[a]) {
            const Pose np = reset_agent(p, a, tm, tm_hbm);
            pos = make_float2(np.x, np.y); d = np.d;
        } else if (act == 1 || act == 2) {                                  // SR:150
            const float2 dv = p.dir_table[d];                               // SR:153
            const float ix = p.inc * dv.x, iy = p.inc * dv.y;
            const float nx = act == 1 ? pos.x + ix : pos.x - ix;            // UT:16-17
            const float ny = act == 1 ? pos.y + iy : pos.y - iy;
            const Collide c = player_colliding(tm, p.H, p.W, nx, ny, p.radius_sq);   // SR:162-163
            if (c.wall == 2 || c.goal == 2) {
                atomicCAS(p.err, 0, RCW_ERR_OUT_OF_BOUNDS);                 // BoundsError: no mutation
            } else if (c.goal) {
                p.reward[a] = p.goal_reward; p.done[a] = 1;                 // SR:166-168
            } else if (c.wall) {
                p.reward[a] = 0.0f; p.done[a] = 0;                          // SR:170-171
            } else {
                pos = make_float2(nx, ny);
                p.pos[a] = pos; p.reward[a] = 0.0f; p.done[a] = 0;          // SR:174-176
            }
        } else if (act == 3 || act == 4) {
            d = act == 3 ? d + 1 : d - 1;                                   // UT:13-14
            d = d >= p.nd ? d - p.nd : (d < 0 ? d + p.nd : d);              // mod(d±1, nd)
            p.dir[a] = d; p.reward[a] = 0.0f; p.done[a] = 0;                // SR:185-187
        }
        s_pose[0] = pos.x; s_pose[1] = pos.y; s_pose[2] = __int_as_float(d);
    }
    __syncthreads();
    const float x = s_pose[0], y = s_pose[1];
    const int d = __builtin_amdgcn_readfirstlane(__float_as_int(s_pose[2]));

    // ---- phase 1: one lane per view column --------------------------------------------------
    const float* tab = p.ray_table + (size_t)d * RCW_TABLE_ROWS * p.N;
    for (int i = tid; i < p.N; i += kBlock) {                               // SR:220, SR:401
        const float dx = tab[i], dy = tab[p.N + i];
        const float ddx = tab[2 * p.N + i], ddy = tab[3 * p.N + i];
        const float dot = tab[4 * p.N + i];
        const RayHit r = cast_ray(tm, p.H, p.W, x, y, dx, dy, ddx, ddy, p.tie_le, p.dist_pre);
        if (r.oob) atomicCAS(p.err, 0, RCW_ERR_OUT_OF_BOUNDS);
        const int h = r.oob ? p.Hc : height_line_pu(p, r.dist, dot);
        // SR:417-429: wall / goal by the WALL bit of the stop tile, shade by hit dimension
        const int cid = ((r.bits & 1u) ? 0 : 2) + (r.dim == 1 ? 0 : 1);
        const int k = p.N - 1 - i;                                          // SR:431 (0-based)
        s_pad[k] = column_padding(p.Hc, h);
        s_col[k] = p.colour[cid];
        if (p.col_h) { p.col_h[(size_t)a * p.N + k] = h; p.col_c[(size_t)a * p.N + k] = (uint8_t)cid; }
    }
    __syncthreads();

    // ---- phase 2: stream the frame (Hc, N) column-major --------------------------------------
    uint32_t* frame = p.obs + (size_t)a * p.N * p.Hc;
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    if (HC256) {
        // one wavefront store instruction = one column: lane l writes rows 4l..4l+3
        const int wave = tid >> 6, lane = tid & 63;
        const int r0 = lane * 4;
        u32x4* out = reinterpret_cast<u32x4*>(frame);
#pragma unroll 4
        for (int k = wave; k < p.N; k += kBlock / 64) {
            const int pad = s_pad[k];
            const uint32_t c = s_col[k];
            u32x4 v;
            v.x = pixel(r0 + 0, pad, 256, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, 256, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, 256, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, 256, c, ceil_c, floor_c);
            __builtin_nontemporal_store(v, out + (size_t)k * 64 + lane);
        }
    } else if ((p.Hc & 3) == 0) {
        const int vpc = p.Hc >> 2;                       // 16-byte vectors per column
        const int total = p.N * vpc;
        u32x4* out = reinterpret_cast<u32x4*>(frame);
        for (int idx = tid; idx < total; idx += kBlock) {
            const int k = idx / vpc;
            const int r0 = (idx - k * vpc) * 4;
            const int pad = s_pad[k];
            const uint32_t c = s_col[k];
            u32x4 v;
            v.x = pixel(r0 + 0, pad, p.Hc, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, p.Hc, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, p.Hc, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, p.Hc, c, ceil_c, floor_c);
            __builtin_nontemporal_store(v, out + idx);
        }
    } else {
        const int total = p.N * p.Hc;
        for (int idx = tid; idx < total; idx += kBlock) {
            const int k = idx / p.Hc;
            const int r = idx - k * p.Hc;
            __builtin_nontemporal_store(pixel(r, s_pad[k], p.Hc, s_col[k], ceil_c, floor_c), frame + idx);
        }
    }
}

// ---- small kernels ---------------------------------------------------------------------------
// @assert action in 1:4  SR:140, for device-resident actions
__global__ void rcw_validate_kernel(const RcwDev p, const uint8_t* __restrict__ actions)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.B) return;
    const int act = actions[a];
    if (act < 1 || act > RCW_NUM_ACTIONS) atomicCAS(p.err, 0, RCW_ERR_INVALID_ACTION);
}

// wall ring SR:57-60 and a placeholder goal at (2,2) (cleared by the first reset)
__global__ void rcw_init_tile_map_kernel(const RcwDev p)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.B) return;
    uint32_t* tm = p.tile_map + (size_t)a * p.nwords;
    for (int w = 0; w < p.nwords; ++w) tm[w] = 0u;
    for (int j = 1; j <= p.W; ++j)
        for (int i = 1; i <= p.H; ++i)
            if (i == 1 || i == p.H || j == 1 || j == p.W) {
                const int t = (i - 1) + p.H * (j - 1);
                tm[t >> 4] |= 1u << ((t & 15) * 2);
            }
    p.goal[a] = make_int2(2, 2);
    p.episode[a] = 0;
    p.reward[a] = 0.0f;
    p.done[a] = 0;
}

__global__ void rcw_reset_kernel(const RcwDev p, const uint8_t* __restrict__ mask)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.B) return;
    if (mask != nullptr && mask[a] == 0) return;
    reset_agent(p, a, p.tile_map + (size_t)a * p.nwords, nullptr);
}

// inject post-reset state: SR:118-132 with caller-chosen draws
__global__ void rcw_set_state_kernel(const RcwDev p, const int2* __restrict__ goal,
                                     const float2* __restrict__ pos, const int32_t* __restrict__ dir,
                                     const uint8_t* __restrict__ mask)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.B) return;
    if (mask != nullptr && mask[a] == 0) return;
    uint32_t* tm = p.tile_map + (size_t)a * p.nwords;
    const int2 old = p.goal[a];
    set_goal_bit(tm, p.H, old.x, old.y, false);   // SR:118
    const int2 g = goal[a];
    p.goal[a] = g;                                // SR:121
    set_goal_bit(tm, p.H, g.x, g.y, true);        // SR:122
    p.pos[a] = pos[a];                            // SR:126
    p.dir[a] = dir[a];                            // SR:129
    p.reward[a] = 0.0f;                           // SR:131
    p.done[a] = 0;                                // SR:132
}

// cast_rays!(world) SR:195-231 with the ray buffers materialised (rcw_rays)
__global__ __launch_bounds__(kBlock) void rcw_rays_kernel(const RcwDev p, int first, RcwRayOut out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int local = blockIdx.x;
    const int a = first + local;
    const int tid = threadIdx.x;
    uint32_t* tm = lds;
    const uint32_t* tm_hbm = p.tile_map + (size_t)a * p.nwords;
    for (int w = tid; w < p.nwords; w += kBlock) tm[w] = tm_hbm[w];
    __syncthreads();
    const float2 pos = p.pos[a];
    const int d = p.dir[a];
    const float* tab = p.ray_table + (size_t)d * RCW_TABLE_ROWS * p.N;
    for (int i = tid; i < p.N; i += kBlock) {
        const float dx = tab[i], dy = tab[p.N + i];
        const RayHit r = cast_ray(tm, p.H, p.W, pos.x, pos.y, dx, dy, tab[2 * p.N + i],
                                  tab[3 * p.N + i], p.tie_le, p.dist_pre);
        const size_t q = (size_t)local * p.N + i;
        if (out.stop_ij) { out.stop_ij[2 * q] = r.oob ? 1 : r.i; out.stop_ij[2 * q + 1] = r.oob ? 1 : r.j; }
        if (out.hit_dim) out.hit_dim[q] = r.oob ? 0 : r.dim;
        if (out.dist) out.dist[q] = r.oob ? 0.0f : r.dist;
        if (out.dirs) { out.dirs[2 * q] = dx; out.dirs[2 * q + 1] = dy; }
    }
}

// receiving side of the compact observation gather: (height_line_pu, colour id) -> frame
__global__ __launch_bounds__(kBlock) void rcw_expand_kernel(const RcwDev p,
                                                            const int32_t* __restrict__ col_h,
                                                            const uint8_t* __restrict__ col_c,
                                                            uint32_t* __restrict__ frames)
{
    const int a = blockIdx.x;
    const int tid = threadIdx.x;
    uint32_t* frame = frames + (size_t)a * p.N * p.Hc;
    const int total = p.N * p.Hc;
    if ((p.Hc & 3) == 0) {
        const int vpc = p.Hc >> 2;
        u32x4* out = reinterpret_cast<u32x4*>(frame);
        for (int idx = tid; idx < (total >> 2); idx += kBlock) {
            const int k = idx / vpc;
            const int r0 = (idx - k * vpc) * 4;
            const int pad = column_padding(p.Hc, col_h[(size_t)a * p.N + k]);
            const uint32_t c = p.colour[col_c[(size_t)a * p.N + k] & 3];
            u32x4 v;
            v.x = pixel(r0 + 0, pad, p.Hc, c, p.ceiling_color, p.floor_color);
            v.y = pixel(r0 + 1, pad, p.Hc, c, p.ceiling_color, p.floor_color);
            v.z = pixel(r0 + 2, pad, p.Hc, c, p.ceiling_color, p.floor_color);
            v.w = pixel(r0 + 3, pad, p.Hc, c, p.ceiling_color, p.floor_color);
            __builtin_nontemporal_store(v, out + idx);
        }
    } else {
        for (int idx = tid; idx < total; idx += kBlock) {
            const int k = idx / p.Hc;
            const int r = idx - k * p.Hc;
            const int pad = column_padding(p.Hc, col_h[(size_t)a * p.N + k]);
            const uint32_t c = p.colour[col_c[(size_t)a * p.N + k] & 3];
            frame[idx] = pixel(r, pad, p.Hc, c, p.ceiling_color, p.floor_color);
        }
    }
}

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------
size_t rcw_step_lds_bytes(const RcwDev& p)
{
    const size_t nwords_pad = ((size_t)p.nwords + 3) & ~(size_t)3;
    return (nwords_pad + 2 * (size_t)p.N) * sizeof(uint32_t);
}

hipError_t rcw_launch_step(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev,
                           hipStream_t s)
{
    const size_t lds = rcw_step_lds_bytes(p);
    if (p.Hc == 256)
        hipLaunchKernelGGL(rcw_step_kernel<true>, dim3(p.B), dim3(kBlock), lds, s, p, actions_dev, mask_dev);
    else
        hipLaunchKernelGGL(rcw_step_kernel<false>, dim3(p.B), dim3(kBlock), lds, s, p, actions_dev, mask_dev);
    return hipGetLastError();
}

hipError_t rcw_launch_validate(const RcwDev& p, const uint8_t* actions_dev, hipStream_t s)
{
    hipLaunchKernelGGL(rcw_validate_kernel, dim3((p.B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, p, actions_dev);
    return hipGetLastError();
}

hipError_t rcw_launch_reset(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s)
{
    hipLaunchKernelGGL(rcw_reset_kernel, dim3((p.B + 63) / 64), dim3(64), 0, s, p, mask_dev);
    return hipGetLastError();
}

hipError_t rcw_launch_set_state(const RcwDev& p, const int2* goal, const float2* pos,
                                const int32_t* dir, const uint8_t* mask_dev, hipStream_t s)
{
    hipLaunchKernelGGL(rcw_set_state_kernel, dim3((p.B + 63) / 64), dim3(64), 0, s, p, goal, pos, dir, mask_dev);
    return hipGetLastError();
}

hipError_t rcw_launch_init_tile_map(const RcwDev& p, hipStream_t s)
{
    hipLaunchKernelGGL(rcw_init_tile_map_kernel, dim3((p.B + 63) / 64), dim3(64), 0, s, p);
    return hipGetLastError();
}

hipError_t rcw_launch_rays(const RcwDev& p, int32_t first, int32_t count, RcwRayOut out, hipStream_t s)
{
    const size_t lds = (((size_t)p.nwords + 3) & ~(size_t)3) * sizeof(uint32_t);
    hipLaunchKernelGGL(rcw_rays_kernel, dim3(count), dim3(kBlock), lds, s, p, first, out);
    return hipGetLastError();
}

hipError_t rcw_launch_expand(const RcwDev& p, const int32_t* col_h, const uint8_t* col_c,
                             int32_t count, uint32_t* frames, hipStream_t s)
{
    hipLaunchKernelGGL(rcw_expand_kernel, dim3(count), dim3(kBlock), 0, s, p, col_h, col_c, frames);
    return hipGetLastError();
}
