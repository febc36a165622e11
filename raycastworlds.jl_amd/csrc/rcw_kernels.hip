// HIP kernels (gfx950 / CDNA4) for the batched SingleRoom step/render path.
//
// A step is two launches on the handle's stream:
//   rcw_cast_kernel   one workgroup per agent.  Its loads go out in two batches, each awaited once: the agent's
//            state (scalar loads in one asm statement + the tile-map words), then what depends on the heading
//            after the action (direction vector, the heading's ray-table entries), issued before the
//            tile_map (2·H·W bits) is unpacked into LDS, a byte per tile, and hidden behind the dynamics;
//            every lane runs the (wave-uniform) dynamics
//            act!(world, a) SR:139-191 redundantly so nothing has to be broadcast — the opt-in
//            re-sample SR:110-137 runs on lane 0 and goes through LDS; then one lane per
//            view column: the ray's table entries (SR:214-221), grid DDA against
//            the LDS tile map (RayCaster.cast_ray, SR:223), perpendicular distance and column
//            height (SR:404-411), colour (SR:417-429) -> a 5-byte column descriptor in HBM,
//            mirrored to image column k = N - i + 1 (SR:431).
//   rcw_fill*_kernel  the bandwidth kernel (update_camera_view!'s column fill SR:431-440):
//            a small fixed grid sweeps one compact window through the (H_cam, N, B) batch,
//            lanes mapped along the image's contiguous axis (rows of one column, Julia
//            column-major), one 16-byte store per lane, one whole 1 KiB column per wavefront
//            store instruction at H_cam = 256 (rcw_fill256_kernel); rcw_fill_window_kernel for
//            256 k / 128 / 64 rows, rcw_fill_flat_kernel for every other height from 24 rows
//            (256-pixel chunks of the flat batch, each lane finds its own column).
// This is an integer/indexing + streaming-store path: no MFMA, the roofline is HBM write
// bandwidth, and the frame (4·H_cam·N bytes per agent-step) is written exactly once.
//
// (opt-in) the reference's other per-step image, update_top_view! SR:446-483, every pixel written once:
//   rcw_fill256_draw_kernel / rcw_top_draw_kernel + rcw_top_store_kernel / rcw_top_store_flat_kernel / rcw_top_store_units_kernel   rays -> lines in
//            an LDS bit plane -> the plane (1/32 of the image) to HBM, in the camera fill's own launch (256-row camera view: the
//            first workgroups fill, the others draw) or as a kernel of its own on a side stream beside the fill kernel; then the fill
//            kernel's moving window over the image with the top view's pixel logic (_flat: any pixel scale from 9 pixels a
//            tile, 256-pixel chunks of the flat batch, descriptor loads one group ahead awaited with vmcnt(63));
//   rcw_top_view_kernel           the same in one persistent kernel (draw and store groups, a ring of LDS planes);
//   rcw_top_view_inplace_kernel   images whose bit plane does not fit in LDS.
//
// Floating point: every operation below is a single IEEE-754 rounding in the world-unit type T
// (Float32, or Float64 for SingleRoom(; T = Float64)), exactly as the reference (Julia never
// contracts a*b+c): this file MUST be compiled with -ffp-contract=off and without fast-math;
// division and sqrt are the correctly rounded forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), denormals are kept.
#include "rcw_kernels.h"
#include "rcw_rng.h"
#include "../../include/rcw.h"

#include <limits.h>
#include <stdlib.h>

#include <mutex>


namespace {

constexpr int kBlock = 256;   // 4 wavefronts of 64
constexpr int kFlatMaxCols = 12;   // image columns a 256-pixel chunk of the flat batch may touch in rcw_fill_flat_kernel (H_cam >= 24; 53 KiB of descriptors in LDS)

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

// ---- the reference's world-unit type T (SingleRoom(; T = ...) SR:259), compiled in: every
// Float32 operation of the path is the same operation in T.  R (the reward type) is independent: store_reward. ----
template <typename T> struct Real;
template <> struct Real<float> {
    typedef float2 vec2;
    static __device__ __forceinline__ vec2* pos(const RcwDev& p) { return p.pos; }
    static __device__ __forceinline__ const vec2* dir_table(const RcwDev& p) { return p.dir_table; }
    static __device__ __forceinline__ const float* ray_table(const RcwDev& p) { return p.ray_table; }
    static __device__ __forceinline__ float radius(const RcwDev& p) { return p.radius; }
    static __device__ __forceinline__ float radius_sq(const RcwDev& p) { return p.radius_sq; }
    static __device__ __forceinline__ float inc(const RcwDev& p) { return p.inc; }
    static __device__ __forceinline__ float num(const RcwDev& p) { return p.num; }
    static __device__ __forceinline__ float two_fov(const RcwDev& p) { return p.two_fov; }
    static __device__ __forceinline__ vec2 make(float x, float y) { return make_float2(x, y); }
};
template <> struct Real<double> {
    typedef double2 vec2;
    static __device__ __forceinline__ vec2* pos(const RcwDev& p) { return p.pos64; }
    static __device__ __forceinline__ const vec2* dir_table(const RcwDev& p) { return p.dir_table64; }
    static __device__ __forceinline__ const double* ray_table(const RcwDev& p) { return p.ray_table64; }
    static __device__ __forceinline__ double radius(const RcwDev& p) { return p.radius64; }
    static __device__ __forceinline__ double radius_sq(const RcwDev& p) { return p.radius_sq64; }
    static __device__ __forceinline__ double inc(const RcwDev& p) { return p.inc64; }
    static __device__ __forceinline__ double num(const RcwDev& p) { return p.num64; }
    static __device__ __forceinline__ double two_fov(const RcwDev& p) { return p.two_fov64; }
    static __device__ __forceinline__ vec2 make(double x, double y) { return make_double2(x, y); }
};
__device__ __forceinline__ float rabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double rabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float rfloor(float x) { return floorf(x); }
__device__ __forceinline__ double rfloor(double x) { return floor(x); }
// floor(Int, x) saturated to Int32 (the reference raises InexactError only beyond Int64)
__device__ __forceinline__ int floor_to_int32(float f)
{
    const int h = (int)fminf(fmaxf(f, -2147483648.0f), 2147483520.0f);
    return f >= 2147483648.0f ? INT_MAX : h;
}
__device__ __forceinline__ int floor_to_int32(double f) { return (int)fmin(fmax(f, -2147483648.0), 2147483647.0); }

// In LDS the tile map is staged UNPACKED, one byte per tile (value = the tile's 2 bits), so a
// lookup in the ray march is a single ds_read_u8 at the linear tile index.
// The LAST tile — the wall ring's corner (H, W), a wall in every map the engine builds — is staged as an obstacle whatever
// HBM holds: cast_ray's march reads it for any index outside the map and relies on it to stop (see there).
__device__ __forceinline__ void stage_tile_bytes(uint8_t* tb, const uint32_t* tm_hbm, int HW, int tid, int nthreads)
{
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int t = tid; t < HW; t += nthreads) {
        const uint32_t b = (tm_hbm[t >> 4] >> ((t & 15) * 2)) & 3u;
        tb[t] = (uint8_t)(t == HW - 1 ? (b | 1u) : b);
    }
}

// ---- is_player_colliding for both layers in one sweep  (CD:21-42) ----------------------
// Per layer the reference walks the 3x3 tiles (j outer, i inner), indexes the map first
// (BoundsError if the tile is off the map) and returns at the first hit.  Result per
// layer: 0 = false, 1 = true, 2 = BoundsError.
struct Collide { int wall, goal; };
// Wave-parallel form: lane t < 9 tests tile t of the neighbourhood in the reference's visiting
// order (t = 3 (j - jt + 1) + (i - it + 1)); three ballots recover "first event in order" per
// layer.  Every lane of the wave gets the same (wave-uniform) result.  px, py are uniform.
template <typename T>
__device__ __forceinline__ Collide player_colliding(const uint8_t* tb, int H, int W, T px, T py, T radius_sq,
                                                    int oob_empty)
{
    const int it = (int)rfloor(px) + 1;   // wu_to_tu UT:5
    const int jt = (int)rfloor(py) + 1;
    const int t = (int)(threadIdx.x & 63u);
    const int tq = t / 3;
    const int i = it - 1 + (t - 3 * tq), j = jt - 1 + tq;
    const bool valid = t < 9;
    const bool inb = i >= 1 && i <= H && j >= 1 && j <= W;
    const uint32_t bits = (valid && inb) ? (uint32_t)tb[(i - 1) + H * (j - 1)] : 0u;
    const T half = (T)0.5;
    const T cx = (T)i - half, cy = (T)j - half;                      // CD:33-34
    const T qx = px - cx, qy = py - cy;                              // CD:35
    const T sx = qx < -half ? -half : (qx > half ? half : qx);       // clamp CD:11
    const T sy = qy < -half ? -half : (qy > half ? half : qy);
    const T vx = qx - sx, vy = qy - sy;                              // CD:16
    const T vx2 = vx * vx, vy2 = vy * vy;
    const bool hit = (vx2 + vy2) < radius_sq;                        // CD:18
    const unsigned long long m_oob = __ballot(valid && !inb && !oob_empty);
    const unsigned long long m_wall = __ballot(hit && (bits & 1u));
    const unsigned long long m_goal = __ballot(hit && (bits & 2u));
    const int first_oob = m_oob ? __builtin_ctzll(m_oob) : 64;
    const int first_wall = m_wall ? __builtin_ctzll(m_wall) : 64;
    const int first_goal = m_goal ? __builtin_ctzll(m_goal) : 64;
    Collide c;
    c.wall = first_wall < first_oob ? 1 : (first_oob < 64 ? 2 : 0);
    c.goal = first_goal < first_oob ? 1 : (first_oob < 64 ? 2 : 0);
    return c;
}

// ---- world.reward::R (SR:33): zero(R) or goal_reward = one(R), stored in the handle's R -----------
__device__ __forceinline__ void store_reward(const RcwDev& p, int a, bool goal)
{
    switch (p.reward_type) {
    case RCW_REWARD_FLOAT64: static_cast<double*>(p.reward)[a] = goal ? p.goal_reward64 : 0.0; break;
    case RCW_REWARD_INT32:   static_cast<int32_t*>(p.reward)[a] = goal ? (int32_t)p.goal_reward64 : 0; break;
    case RCW_REWARD_INT64:   static_cast<int64_t*>(p.reward)[a] = goal ? (int64_t)p.goal_reward64 : 0; break;
    default:                 static_cast<float*>(p.reward)[a] = goal ? p.goal_reward : 0.0f; break;
    }
}

// ---- reset!(world)  SR:110-137 with the counter-based generator -------------------------
// tm_a / tm_b: the agent's tile map words in up to two places (LDS copy and HBM).
template <typename T> struct Pose { T x, y; int d; };
template <typename T>
__device__ __forceinline__ Pose<T> reset_agent(const RcwDev& p, int a, uint32_t* tm_a, uint32_t* tm_b)
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
    bool gave_up = true;
    for (uint64_t t = 0; t < max_tries; ++t) {                                 // UT:26
        const int ti = (int)(lin % (uint64_t)H) + 1, tj = (int)(lin / (uint64_t)H) + 1;
        if (tile_bits(tm_a, H, ti, tj)) lin = rcw_below(rcw_draw(key, n++), HW);   // UT:27-28
        else { gave_up = false; break; }
    }
    // UT:34: "@warn Could not sample an empty position in max_tries ... Returning non-empty position" — the reference goes on
    // with the occupied tile; so does the engine, and says so in the agent's status word (a warning: no error word, no call fails)
    if (gave_up && p.status[a] == 0) p.status[a] = RCW_WARN_SAMPLER_GAVE_UP;
    const int pi = (int)(lin % (uint64_t)H) + 1, pj = (int)(lin / (uint64_t)H) + 1;
    Pose<T> o;
    o.x = (T)((double)pi - 0.5);                                              // SR:125
    o.y = (T)((double)pj - 0.5);
    o.d = (int)rcw_below(rcw_draw(key, n++), (uint64_t)p.nd);                  // SR:128
    Real<T>::pos(p)[a] = Real<T>::make(o.x, o.y);                             // SR:126
    p.dir[a] = o.d;                                                           // SR:129
    store_reward(p, a, false);                                                // SR:131
    p.done[a] = 0;                                                            // SR:132
    p.episode[a] = ep + 1;
    return o;
}

// ---- RayCaster.cast_ray  (external; call site SR:223).  UNPINNED choices via p.tie_le /
// p.dist_pre (include/rcw.h).  Leaves the map -> oob (Julia: BoundsError). ------------------
template <typename T> struct RayHit { int t, dim; T dist; uint32_t bits; bool oob; };
// The march is written with selects, not branches (lanes of a wavefront disagree on the step
// axis at almost every iteration; a divergent if/else costs more in exec-mask bookkeeping than
// the few v_cndmask), and it carries only what the result needs: the two side distances, the
// linear tile index t (the stop tile is t mod H, t div H) and the last step.  The chip's
// instruction issue bounds this loop (B·N rays x trip count x instructions — scalar ones count like
// vector ones), so every instruction counts: ONE exit condition (the tile's byte), nothing counted.
// Termination needs no counter: t moves strictly monotonically along both axes, an index outside
// [0, H·W) reads the LAST tile instead, and stage_tile_bytes makes that byte an obstacle whatever
// HBM holds (it is the wall ring's corner, SR:57-60: a wall in every map the engine builds) — so a
// corrupt map ends a ray at the latest when it leaves the index range, reported as out of bounds.
// The hit dimension is read off the last step (+-1 along x, +-H along y; H >= 3).
template <typename T, bool TIE_LE, bool DIST_PRE>
__device__ __forceinline__ RayHit<T> cast_ray(const uint8_t* tb, int H, int W, T x, T y, T dx, T dy, T ddx, T ddy)
{
    const int i0 = (int)rfloor(x) + 1;    // wu_to_tu UT:5
    const int j0 = (int)rfloor(y) + 1;
    const bool neg_x = dx < (T)0, neg_y = dy < (T)0;
    const int si = neg_x ? -1 : 1;
    const int tj = neg_y ? -H : H;
    const T fx = neg_x ? x - (T)(i0 - 1) : (T)i0 - x;
    const T fy = neg_y ? y - (T)(j0 - 1) : (T)j0 - y;
    T sx = fx * ddx, sy = fy * ddy;
    int t = (i0 - 1) + H * (j0 - 1);
    const unsigned last = (unsigned)(H * W - 1);
    RayHit<T> r;
    r.dist = (T)0;
    int step = 0;                                                             // the last step: si, tj, or none
    r.bits = tb[(unsigned)t < last ? (unsigned)t : last];                     // never read outside the map
    while (r.bits == 0u) {
        const bool xf = TIE_LE ? (sx <= sy) : (sx < sy);
        const T nx = sx + ddx, ny = sy + ddy;
        if (DIST_PRE) r.dist = xf ? sx : sy;
        sx = xf ? nx : sx;
        sy = xf ? sy : ny;
        step = xf ? si : tj;
        t += step;
        r.bits = tb[(unsigned)t < last ? (unsigned)t : last];
    }
    r.dim = step == 0 ? 0 : (step == si ? 1 : 2);
    r.oob = (unsigned)t > last;
    if (!DIST_PRE) {
        const T d1 = sx - ddx, d2 = sy - ddy;
        r.dist = r.dim == 1 ? d1 : (r.dim == 2 ? d2 : (T)0);
    }
    r.t = t;
    return r;
}

// The same march for the cast kernel, which lays the tile bytes out with a GUARD BAND of H obstacle bytes in front of
// tile 0 and behind the last tile: a step moves the linear index by 1 or by H, so the first index outside the map falls
// into a band, reads as an obstacle and ends the ray — no clamp in the loop.  The loop carries the LDS byte ADDRESS of
// the current tile (tile index + the array's LDS address, added once) and reads it with ds_read_u8 by name: written in C
// the compiler re-adds the (link-time) base of the dynamic LDS array to the index in every iteration.  8 vector
// instructions and the LDS read per tile crossed (the clamped form: 10; the round-2 form with its step counter: 14 and
// eleven scalar ones) — at the deep-march config (32×32 map, rays of up to 60 tiles) the kernel is issue-bound.
template <typename T, bool TIE_LE, bool DIST_PRE>
__device__ __forceinline__ RayHit<T> cast_ray_guarded(const uint8_t* tiles, int H, int W, T x, T y, T dx, T dy, T ddx, T ddy)
{
    const int i0 = (int)rfloor(x) + 1;    // wu_to_tu UT:5
    const int j0 = (int)rfloor(y) + 1;
    const bool neg_x = dx < (T)0, neg_y = dy < (T)0;
    const int si = neg_x ? -1 : 1;
    const int tj = neg_y ? -H : H;
    const T fx = neg_x ? x - (T)(i0 - 1) : (T)i0 - x;
    const T fy = neg_y ? y - (T)(j0 - 1) : (T)j0 - y;
    T sx = fx * ddx, sy = fy * ddy;
    const uint32_t base = (uint32_t)reinterpret_cast<size_t>((__attribute__((address_space(3))) const uint8_t*)tiles);   // LDS address of tile 0
    const int t0 = (i0 - 1) + H * (j0 - 1);
    const unsigned last = (unsigned)(H * W - 1);
    // (a start outside the map — no state the engine produces — starts in the front band: the march ends at once)
    uint32_t u = base + ((unsigned)t0 <= last ? (uint32_t)t0 : 0xFFFFFFFFu);
    RayHit<T> r;
    r.dist = (T)0;
    int step = 0;                                                             // the last step: si, tj, or none
    asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r.bits) : "v"(u) : "memory");
    while (r.bits == 0u) {
        const bool xf = TIE_LE ? (sx <= sy) : (sx < sy);
        const T nx = sx + ddx, ny = sy + ddy;
        if (DIST_PRE) r.dist = xf ? sx : sy;
        sx = xf ? nx : sx;
        sy = xf ? sy : ny;
        step = xf ? si : tj;
        u += (uint32_t)step;
        asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r.bits) : "v"(u) : "memory");
    }
    const int t = (int)(u - base);
    r.dim = step == 0 ? 0 : (step == si ? 1 : 2);
    r.oob = (unsigned)t > last;
    if (!DIST_PRE) {
        const T d1 = sx - ddx, d2 = sy - ddy;
        r.dist = r.dim == 1 ? d1 : (r.dim == 2 ? d2 : (T)0);
    }
    r.t = t;
    return r;
}

#ifdef RCW_DEV_SWITCHES
// The march north_star words literally: a wave-uniform loop bound resolved by ballot.  Every lane of the wavefront
// iterates until the LAST ray of the wavefront has hit (`__ballot(still marching) != 0`), finished lanes carrying their
// state through selects, so there is no divergent branch at all.  Same results as cast_ray, bit for bit; kept as a
// measured alternative (cfg.-independent development switch RCW_CAST_MARCH=ballot, profiles/r02_cast_march_cfg5.txt):
// the exec-masked `break` of cast_ray runs the same number of iterations (the wavefront's longest ray) with fewer
// instructions per iteration, because the hardware's exec mask does for free what the selects do explicitly.
template <typename T, bool TIE_LE, bool DIST_PRE>
__device__ __forceinline__ RayHit<T> cast_ray_ballot(const uint8_t* tb, int H, int W, T x, T y, T dx, T dy, T ddx, T ddy)
{
    const int i0 = (int)rfloor(x) + 1;
    const int j0 = (int)rfloor(y) + 1;
    const bool neg_x = dx < (T)0, neg_y = dy < (T)0;
    const int si = neg_x ? -1 : 1;
    const int tj = neg_y ? -H : H;
    const T fx = neg_x ? x - (T)(i0 - 1) : (T)i0 - x;
    const T fy = neg_y ? y - (T)(j0 - 1) : (T)j0 - y;
    T sx = fx * ddx, sy = fy * ddy;
    int t = (i0 - 1) + H * (j0 - 1);
    const unsigned last = (unsigned)(H * W - 1);
    const int cap = H + W;
    RayHit<T> r;
    r.dim = 0; r.dist = (T)0; r.bits = 0u;
    int n = 0;
    bool marching = true;
    while (__ballot(marching) != 0ull) {
        const unsigned tc = (unsigned)t < last ? (unsigned)t : last;
        const uint32_t bits = tb[tc];
        const bool stop = bits != 0u || n >= cap;
        if (marching) r.bits = bits;
        marching = marching && !stop;
        n += marching ? 1 : 0;
        const bool xf = TIE_LE ? (sx <= sy) : (sx < sy);
        const T nx = sx + ddx, ny = sy + ddy;
        if (DIST_PRE) r.dist = marching ? (xf ? sx : sy) : r.dist;
        sx = (marching && xf) ? nx : sx;
        sy = (marching && !xf) ? ny : sy;
        t += marching ? (xf ? si : tj) : 0;
        r.dim = marching ? (xf ? 1 : 2) : r.dim;
    }
    r.oob = r.bits == 0u || (unsigned)t > last;
    if (!DIST_PRE) {
        const T d1 = sx - ddx, d2 = sy - ddy;
        r.dist = r.dim == 1 ? d1 : (r.dim == 2 ? d2 : (T)0);
    }
    r.t = t;
    return r;
}
#endif   // RCW_DEV_SWITCHES

// ---- column height  SR:404-411 ------------------------------------------------------------
template <typename T>
__device__ __forceinline__ int height_line_pu(const RcwDev& p, T dist, T dot)
{
    const T projected = dist * dot;                           // SR:404
    const T den = Real<T>::two_fov(p) * projected;            // (2 * fov) * projected
    const T height_line = Real<T>::num(p) / den;              // SR:406
    const int h = floor_to_int32(rfloor(height_line));        // floor(Int, .) SR:408
    return isfinite(height_line) ? h : p.Hc;                  // SR:407-411
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

// floor(n / d) for n >= 0, d >= 1 without the integer-division sequence: the Float32 quotient is off by at most one,
// which the two corrections repair.  Preconditions: (q + 1)·d fits int32, i.e. n < 2^31 - d, and the error of
// (float)n · (1/d) stays below one — certain for n < 2^23 (everything is then exact to a rounding), and also for larger n
// as long as the QUOTIENT is small: the relative error is ~2^-22, so n / d <= 2^13 keeps it below 2^-9.  The callers:
// rcw_fill_flat_kernel / rcw_top_store_flat_kernel (n < 2^20 + 256), top_store (n < 2^14) and rcw_fill_frame_kernel, whose
// flat index reaches N·H_cam < 2^25 with a quotient (the column) <= N <= 8192 — rcw_launch_fill's guard, restated here
// because widening it would silently produce wrong columns (tests/test_host_logic.py checks the admitted range).
__device__ __forceinline__ int fast_div(int n, int d, float inv_d)
{
    int q = (int)((float)n * inv_d);
    q -= (q * d > n) ? 1 : 0;
    q += ((q + 1) * d <= n) ? 1 : 0;
    return q;
}

#ifdef RCW_DEV_SWITCHES
// ---- the ROUND-3 form of kernel 1 (development build only: RCW_CAST_KERNEL=r3, and the carrier of the two measured-and-rejected
// variants RCW_CAST_MARCH=ballot / RCW_CAST_TABLE=lds).  Its loads are written "up front" but the compiler serialises them: the
// tile-map staging loop waits for its own loads before the action byte is requested, the pose and heading come after that,
// the heading's direction vector after those, the ray-table row after the dynamics — five dependent round trips.  rcw_cast_kernel
// below issues them in two. --------------------------------
// One workgroup per agent.  Output: the agent's new state and one compact descriptor per
// image column (height_line_pu, colour id) — 5 bytes per column, against the 4·H_cam bytes
// of pixels the fill kernel then writes for it.
// TIE_LE / DIST_PRE: the UNPINNED cast_ray choices (include/rcw.h), compiled in.
template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_cast_kernel_r3(const RcwDev p,
                                                          const uint8_t* __restrict__ actions,
                                                          const uint8_t* __restrict__ mask, int first)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int a = first + (int)blockIdx.x;                  // agents [first, first + gridDim.x)
    const int tid = threadIdx.x;
    if (mask != nullptr && mask[a] == 0) return;

    // LDS: [H guard bytes | H*W tile bytes, the agent's tile map | H guard bytes] (cast_ray_guarded) | the re-sampled pose
    typedef typename Real<T>::vec2 vec2;
    const int HW = p.H * p.W;
    uint8_t* const tb = reinterpret_cast<uint8_t*>(lds) + p.H;
    for (int k = tid; k < 2 * p.H; k += (int)blockDim.x) (k < p.H ? tb - p.H + k : tb + HW + (k - p.H))[0] = 1;
    T* const s_pose = reinterpret_cast<T*>(lds + ((HW + 2 * p.H + 15) / 16) * 4);     // [2] + the heading (auto-reset)
    int& s_pose_d = *reinterpret_cast<int*>(s_pose + 2);

    // ---- every load of the agent's state is issued up front (all wave-uniform addresses) ----
    uint32_t* tm_hbm = p.tile_map + (size_t)a * p.nwords;
    stage_tile_bytes(tb, tm_hbm, HW, tid, (int)blockDim.x);
    int act = actions ? (int)actions[a] : 0;
    const vec2 pos = Real<T>::pos(p)[a];
    const int d = p.dir[a];
    const int was_done = p.done[a];
    const vec2 dv = Real<T>::dir_table(p)[d];                               // SR:153
    const bool invalid = actions != nullptr && (act < 1 || act > RCW_NUM_ACTIONS);   // @assert SR:140
    if (invalid) act = 0;                                                   // this agent is not stepped
    const bool resample = act != 0 && p.auto_reset != 0 && was_done != 0;
    // the heading after the action depends on nothing else, so the ray-table row is known now
    int d_new = d;
    if (!resample && act == 3) d_new = d + 1 >= p.nd ? 0 : d + 1;          // turn_left  UT:13
    if (!resample && act == 4) d_new = d - 1 < 0 ? p.nd - 1 : d - 1;       // turn_right UT:14
    // All four wavefronts must have READ the state before lane 0 overwrites it below: pin the
    // loaded values in registers ahead of the barrier.
    asm volatile("" :: "v"(pos.x), "v"(pos.y), "v"(d), "v"(was_done), "v"(act), "v"(dv.x), "v"(dv.y));
    __syncthreads();

    // ---- phase 0: dynamics, computed redundantly by every lane (no broadcast needed) --------
    T x = pos.x, y = pos.y;
    if (resample) {                                                         // wave-uniform, rare
        if (tid == 0) {
            const Pose<T> np = reset_agent<T>(p, a, tm_hbm, nullptr);
            s_pose[0] = np.x; s_pose[1] = np.y; s_pose_d = np.d;
        }
        __syncthreads();
        stage_tile_bytes(tb, tm_hbm, HW, tid, (int)blockDim.x);             // the goal moved
        __syncthreads();
        x = s_pose[0]; y = s_pose[1];
        d_new = s_pose_d;
    } else if (act != 0) {
        int done = 0;                                                       // reward = done ? goal_reward : zero(R)
        bool oob = false;
        if (act <= 2) {                                                     // SR:150
            const T ix = Real<T>::inc(p) * dv.x, iy = Real<T>::inc(p) * dv.y;
            const T nx = act == 1 ? pos.x + ix : pos.x - ix;                // UT:16-17
            const T ny = act == 1 ? pos.y + iy : pos.y - iy;
            const Collide c = player_colliding<T>(tb, p.H, p.W, nx, ny, Real<T>::radius_sq(p), p.oob_empty);   // SR:162-163
            if (c.wall == 2 || c.goal == 2) oob = true;                     // BoundsError: no mutation
            else if (c.goal) { done = 1; }                                  // SR:166-168
            else if (c.wall) { }                                            // SR:170-171
            else { x = nx; y = ny; }                                        // SR:174
        }
        if (tid == 0) {
            if (oob) {
                p.err[0] = RCW_ERR_OUT_OF_BOUNDS;
                p.status[a] = RCW_ERR_OUT_OF_BOUNDS;
            } else {
                Real<T>::pos(p)[a] = Real<T>::make(x, y);                   // SR:174
                p.dir[a] = d_new;                                           // SR:185
                store_reward(p, a, done != 0); p.done[a] = (uint8_t)done;   // SR:167-176, SR:186-187
            }
        }
    }
    if (invalid && tid == 0) { p.err[0] = RCW_ERR_INVALID_ACTION; p.status[a] = RCW_ERR_INVALID_ACTION; }
    d_new = __builtin_amdgcn_readfirstlane(d_new);

    // ---- phase 1: one lane per view column --------------------------------------------------
    const T* tab = Real<T>::ray_table(p) + (size_t)d_new * RCW_TABLE_ROWS * p.N;
#ifdef RCW_DEV_SWITCHES
    if (p.cast_table_lds) {
        // Development switch RCW_CAST_TABLE=lds: stage the heading's table slice (5 N values) in LDS first, as
        // north_star words it, then read it back.  Every entry is used exactly once by exactly one lane, so the copy
        // buys no reuse — measured against the direct, coalesced L2 read below (profiles/).
        T* stab = reinterpret_cast<T*>(lds + ((HW + 2 * p.H + 15) / 16) * 4 + 8);
        __syncthreads();
        for (int k = tid; k < RCW_TABLE_ROWS * p.N; k += (int)blockDim.x) stab[k] = tab[k];
        __syncthreads();
        tab = stab;
    }
#endif
    for (int i = tid; i < p.N; i += (int)blockDim.x) {                        // SR:220, SR:401
        const T dx = tab[i], dy = tab[p.N + i];
        const T ddx = tab[2 * p.N + i], ddy = tab[3 * p.N + i];
        const T dot = tab[4 * p.N + i];
#ifdef RCW_DEV_SWITCHES
        const RayHit<T> r = p.cast_ballot ? cast_ray_ballot<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, x, y, dx, dy, ddx, ddy)
                                          : cast_ray_guarded<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, x, y, dx, dy, ddx, ddy);
#else
        const RayHit<T> r = cast_ray_guarded<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, x, y, dx, dy, ddx, ddy);
#endif
        if (r.oob) { p.err[0] = RCW_ERR_OUT_OF_BOUNDS; p.status[a] = RCW_ERR_OUT_OF_BOUNDS; }
        const int h = r.oob ? p.Hc : height_line_pu<T>(p, r.dist, dot);
        // SR:417-429: wall / goal by the WALL bit of the stop tile, shade by hit dimension
        const int cid = ((r.bits & 1u) ? 0 : 2) + (r.dim == 1 ? 0 : 1);
        const int k = p.N - 1 - i;                                          // SR:431 (0-based)
        p.col_h[(size_t)a * p.N + k] = h;
        p.col_c[(size_t)a * p.N + k] = (uint8_t)cid;
    }
}
#endif   // RCW_DEV_SWITCHES (rcw_cast_kernel_r3)

// ---- kernel 1 of a step: dynamics + ray cast + projection --------------------------------
// One workgroup per agent.  Output: the agent's new state and one compact descriptor per
// image column (height_line_pu, colour id) — 5 bytes per column, against the 4·H_cam bytes
// of pixels the fill kernel then writes for it.
// TIE_LE / DIST_PRE: the UNPINNED cast_ray choices (include/rcw.h), compiled in.
//
// The kernel is bound by its chain of dependent memory round trips and by its instruction count together (a wavefront lives
// ~6 us at cfg-2, most of it waiting: profiles/r03_cast_cfg2_sq.txt), so the loads are arranged in TWO batches, each issued
// back to back and awaited once:
//   1  mask byte, action byte, pose, heading, done flag and this lane's tile-map words (nothing depends on anything);
//   2  what depends on the heading AFTER the action — known as soon as batch 1 is back: turn_left / turn_right touch
//      nothing else (UT:13-14) —: the old heading's direction vector (move_forward / move_backward, UT:16-17) and the new
//      heading's ray-table entries of this lane's first kCastCols view columns (20 registers), in flight while the tile
//      bytes are unpacked into LDS and the dynamics run.
// (The round-3 kernel asked for the same loads in the same order of SOURCE lines; its ISA waited five times:
// rcw_cast_kernel_r3 above, kept in the development build for the comparison.)
// Batch 1 of the cast kernel's loads — mask byte, action byte, done flag, heading, pose: five wave-uniform addresses, five
// SCALAR loads issued back to back and awaited ONCE.  Written as one asm statement because the compiler, left to itself,
// puts each load's first use (a shift, a compare) right behind it and therefore a `s_waitcnt lgkmcnt(0)` after every single
// load (scalar loads return out of order: the counter can only be waited to zero) — five round trips instead of one; as
// vector loads of a uniform address it follows each with v_readfirstlane, with the same effect.
// gfx9 has no scalar byte load: a byte comes as the aligned 32-bit word that holds it (the hardware drops the address's two
// low bits); the word never leaves the byte's page, so it is readable whenever the byte is, and the other three bytes —
// neighbouring agents' — are discarded.  The statement ends with the wait, so nothing is in flight when it returns.
typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
struct CastState { uint32_t mask_w, act_w, done_w; int d; };
__device__ __forceinline__ CastState load_cast_state(const uint8_t* mask_q, const uint8_t* act_q, const uint8_t* done_q, const int32_t* dir_q,
                                                     const float2* pos_q, float2& pos)
{
    CastState c; uint64_t pw;
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %6, 0x0\n\ts_load_dword %2, %7, 0x0\n\ts_load_dword %3, %8, 0x0\n\t"
                 "s_load_dwordx2 %4, %9, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(c.mask_w), "=&s"(c.act_w), "=&s"(c.done_w), "=&s"(c.d), "=&s"(pw)
                 : "s"(mask_q), "s"(act_q), "s"(done_q), "s"(dir_q), "s"(pos_q) : "memory");
    pos.x = __uint_as_float((uint32_t)pw); pos.y = __uint_as_float((uint32_t)(pw >> 32));
    return c;
}
__device__ __forceinline__ CastState load_cast_state(const uint8_t* mask_q, const uint8_t* act_q, const uint8_t* done_q, const int32_t* dir_q,
                                                     const double2* pos_q, double2& pos)
{
    CastState c; su32x4 pw;
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %6, 0x0\n\ts_load_dword %2, %7, 0x0\n\ts_load_dword %3, %8, 0x0\n\t"
                 "s_load_dwordx4 %4, %9, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(c.mask_w), "=&s"(c.act_w), "=&s"(c.done_w), "=&s"(c.d), "=&s"(pw)
                 : "s"(mask_q), "s"(act_q), "s"(done_q), "s"(dir_q), "s"(pos_q) : "memory");
    pos.x = __longlong_as_double((long long)(((uint64_t)pw.y << 32) | pw.x)); pos.y = __longlong_as_double((long long)(((uint64_t)pw.w << 32) | pw.z));
    return c;
}
__device__ __forceinline__ int byte_of_word(uint32_t w, const uint8_t* q) { return (int)((w >> (8u * (uint32_t)(reinterpret_cast<uintptr_t>(q) & 3u))) & 0xffu); }
// base[byte_offset] with a 32-bit byte offset: the uniform base stays in scalar registers and the lane's part of the address
// is one register (global_load ... v_off, s[base]); indexed in C the offset is sign-extended and the address built per lane in
// 64 bits — two more registers and two more instructions for every load.
template <typename T>
__device__ __forceinline__ T load_at(const T* base, uint32_t byte_offset)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_offset);
}

#ifdef RCW_TRACE_WAVES
// Measurement build only (make trace, tools/cast_trace.py): the first wavefront of each of the first 4096 workgroups of
// rcw_cast_kernel leaves s_memrealtime (100 MHz) at eight points of its life, and where it ran.
__device__ unsigned long long g_cast_trace[4096 * 10];
}  // namespace
extern "C" __attribute__((visibility("default"))) int rcw_cast_trace_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cast_trace), sizeof(unsigned long long) * 4096 * 10);
}
namespace {
#define RCW_CAST_STAMP(k) do { if (tid == 0 && trace_slot < 4096) g_cast_trace[trace_slot * 10 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RCW_CAST_STAMP(k) do { } while (0)
#endif

constexpr int kCastCols = 4;    // view columns a lane keeps in registers (cast_block is chosen so that a lane has at most 4)
constexpr int kCastTiles = 4;   // tiles a lane unpacks from words requested in batch 1 (a 32x32 map on 256 lanes); larger maps loop

// One view column: march, projection, descriptor.  Returns whether the ray left the map (the caller reports it once per lane:
// a branch around two stores in every column costs the issue-bound kernel eight instructions a column).  The descriptor arrays
// are addressed as uniform base (the agent's row) + 32-bit lane offset.
template <typename T, bool TIE_LE, bool DIST_PRE, bool PUBLISH = false>
__device__ __forceinline__ bool cast_column(const RcwDev& p, const uint8_t* tb, int32_t* col_h_a, uint8_t* col_c_a, int i, T x, T y, T dx, T dy, T ddx, T ddy, T dot,
                                            uint32_t* hc_a = nullptr)
{
    const RayHit<T> r = cast_ray_guarded<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, x, y, dx, dy, ddx, ddy);
    const int hl = height_line_pu<T>(p, r.dist, dot);
    const int h = r.oob ? p.Hc : hl;
    // SR:417-429: wall / goal by the WALL bit of the stop tile, shade by hit dimension
    const int cid = ((r.bits & 1u) ? 0 : 2) + (r.dim == 1 ? 0 : 1);
    const uint32_t k = (uint32_t)(p.N - 1 - i);                             // SR:431 (0-based)
    *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(col_h_a) + k * 4u) = h;
    *(col_c_a + k) = (uint8_t)cid;
#ifdef RCW_DEV_SWITCHES
    // (rcw_step256_kernel: the two in one word for the fill workgroups of the SAME launch — a write-through store, agent scope)
    if (PUBLISH) __hip_atomic_store(hc_a + k, (uint32_t)column_padding(256, h) | ((uint32_t)cid << 9) | (p.step_epoch << 11), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return r.oob;
}

// ---- the one-launch step's successors (rcw_fill256_cast_kernel) ----------------------------------------------------------
// The word the fill reads for a column of a 256-row camera view: its padding (SR:436, 0..256) | colour id << 9.
__device__ __forceinline__ uint32_t spec_word(int h, int cid) { return (uint32_t)column_padding(256, h) | ((uint32_t)cid << 9); }

// Under the HBM-bound fill every vector-memory operation of the casting workgroups costs the launch several times what it costs alone,
// every vector instruction next to nothing (profiles/r06_step_forms.txt): the casting half therefore LOADS only the ray's direction
// (rows 0, 1 of the heading's table slice) and recomputes the other three entries exactly as the host's table builder made them
// (rcw_api.hip, build_ray_table): |1 / dx|, |1 / dy| — IEEE division, correctly rounded on both sides — and sum(dir .* ray) SR:404 =
// fl(fl(d1 r1) + fl(d2 r2)), one rounding an operation (this file is compiled without contraction).
template <typename T>
__device__ __forceinline__ void spec_derive(T dx, T dy, T hx, T hy, T& ddx, T& ddy, T& dot)
{
    ddx = rabs((T)1 / dx);
    ddy = rabs((T)1 / dy);
    const T m1 = hx * dx, m2 = hy * dy;
    dot = m1 + m2;
}
// a heading's ray directions of this lane's first kCastCols view columns: the loads ...
template <typename T>
__device__ __forceinline__ void spec_load_rows(const T* tab, int tid, int nthr, int N, T* r_dx, T* r_dy)
{
#pragma unroll
    for (int k = 0; k < kCastCols; ++k) {
        const int i = tid + k * nthr;
        const uint32_t o = (uint32_t)(i < N ? i : N - 1) * (uint32_t)sizeof(T);
        r_dx[k] = load_at(tab, o); r_dy[k] = load_at(tab + N, o);
    }
}
// ... and the entries derived from them (hx, hy: the heading's direction vector, directions_wu[d] SR:65-69)
template <typename T>
__device__ __forceinline__ void spec_derive_rows(T hx, T hy, const T* r_dx, const T* r_dy, T* r_ddx, T* r_ddy, T* r_dot)
{
#pragma unroll
    for (int k = 0; k < kCastCols; ++k) spec_derive<T>(r_dx[k], r_dy[k], hx, hy, r_ddx[k], r_ddy[k], r_dot[k]);
}

// One state's whole fan (cast_rays! SR:195-231 + the column of update_camera_view! SR:401-429) from pose (x, y) with the table entries
// of its heading: the packed word of every column into the slots `slots` names (bit s: slot s of the agent's [5][N] words — the five
// slots of an agent lie together, [B][5][N]: what the casting workgroups write is ONE stream through memory beside the fill's);
// COLS && col_h_a: also the (height_line_pu, colour id) descriptors of the current frame, as cast_column.  hx, hy: the heading's
// direction vector (for the columns beyond kCastCols a lane, whose entries are loaded and derived here).  Returns whether a ray left the map.
template <typename T, bool TIE_LE, bool DIST_PRE, bool COLS>
__device__ __forceinline__ bool spec_fan(const RcwDev& p, const uint8_t* tb, int tid, int nthr, T x, T y,
                                         const T* r_dx, const T* r_dy, const T* r_ddx, const T* r_ddy, const T* r_dot, const T* tab, T hx, T hy,
                                         int32_t* col_h_a, uint8_t* col_c_a, uint16_t* slot_a, uint32_t stride, uint32_t slots)
{
    const int N = p.N;
    bool left = false;
    auto column = [&](int i, T dx, T dy, T ddx, T ddy, T dot) {
        const RayHit<T> r = cast_ray_guarded<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, x, y, dx, dy, ddx, ddy);
        const int hl = height_line_pu<T>(p, r.dist, dot);
        const int h = r.oob ? p.Hc : hl;
        const int cid = ((r.bits & 1u) ? 0 : 2) + (r.dim == 1 ? 0 : 1);     // SR:417-429
        const uint32_t k = (uint32_t)(N - 1 - i);                           // SR:431 (0-based)
#ifdef RCW_DEV_SWITCHES
        if (COLS && col_h_a != nullptr && !(p.spec_debug & 4)) {
#else
        if (COLS && col_h_a != nullptr) {                                    // (wave-uniform)
#endif
            *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(col_h_a) + k * 4u) = h;
            *(col_c_a + k) = (uint8_t)cid;
        }
        const uint16_t w = (uint16_t)spec_word(h, cid);
        uint16_t* const q = slot_a + k;
#ifdef RCW_DEV_SWITCHES
        if (p.spec_debug & 4) { asm volatile("" :: "v"(w)); left |= r.oob; return; }   // (timing probe: no slot stores)
#endif
#pragma unroll
        for (int s = 0; s < 5; ++s) if (slots & (1u << s)) q[(uint32_t)s * stride] = w;   // (wave-uniform)
        left |= r.oob;
    };
#pragma unroll
    for (int k = 0; k < kCastCols; ++k) {
        const int i = tid + k * nthr;
        if (i < N) column(i, r_dx[k], r_dy[k], r_ddx[k], r_ddy[k], r_dot[k]);
    }
    if (N > kCastCols * nthr) {                                             // more than kCastCols columns a lane
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int i = tid + kCastCols * nthr; i < N; i += nthr) {
            const T dx = tab[i], dy = tab[N + i];
            T ddx, ddy, dot;
            spec_derive<T>(dx, dy, hx, hy, ddx, ddy, dot);
            column(i, dx, dy, ddx, ddy, dot);
        }
    }
    return left;
}

// reset!(world) SR:110-137 for an agent that is done, WITHOUT committing it: the draws reset_agent will make when the next launch
// re-samples the agent (the generator is a pure function of seed, global agent id, episode and draw index), on the agent's tile BYTES in
// LDS — which it leaves as the re-sampled world's: the goal bit moved (SR:118-122).  The caller casts the new pose against them.
template <typename T>
__device__ __forceinline__ Pose<T> reset_preview(const RcwDev& p, int a, uint8_t* tb)
{
    const int H = p.H, W = p.W;
    const uint64_t key = rcw_episode_key(p.seed, (uint64_t)(p.agent_id_offset + a), (uint64_t)p.episode[a]);
    uint64_t n = 0;
    const int2 old = p.goal[a];
    tb[(old.x - 1) + H * (old.y - 1)] &= (uint8_t)~2u;                      // SR:118
    const int gi = 2 + (int)rcw_below(rcw_draw(key, n++), (uint64_t)(H - 2));  // SR:120
    const int gj = 2 + (int)rcw_below(rcw_draw(key, n++), (uint64_t)(W - 2));
    tb[(gi - 1) + H * (gj - 1)] |= 2u;                                      // SR:122
    const uint64_t HW = (uint64_t)H * (uint64_t)W;
    const uint64_t max_tries = 1024ull * HW;
    uint64_t lin = rcw_below(rcw_draw(key, n++), HW);                        // UT:24
    for (uint64_t t = 0; t < max_tries; ++t) {                               // UT:26 (tile (i, j) is byte (i-1) + H (j-1) = lin)
        if (tb[lin]) lin = rcw_below(rcw_draw(key, n++), HW);                // UT:27-28
        else break;
    }
    const int pi = (int)(lin % (uint64_t)H) + 1, pj = (int)(lin / (uint64_t)H) + 1;
    Pose<T> o;
    o.x = (T)((double)pi - 0.5);                                            // SR:125
    o.y = (T)((double)pj - 0.5);
    o.d = (int)rcw_below(rcw_draw(key, n++), (uint64_t)p.nd);                // SR:128
    return o;
}

// The lanes of ONE agent wait for each other's LDS writes: a workgroup barrier — or, where the agent is a single wavefront
// (WAVE), nothing but the wavefront's own LDS counter: its LDS operations execute in order.
template <bool WAVE>
__device__ __forceinline__ void agent_sync()
{
    if (WAVE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
}

// WAVE = false: the workgroup is one agent (tid = its thread, nthr = blockDim).  WAVE = true (development build only, measured and
// rejected): 64 lanes are an agent and the workgroup's wavefronts are DIFFERENT agents (rcw_cast_waves_kernel): the same code with
// tid = the lane, nthr = 64, the wavefront's own slice of LDS, and no workgroup barrier.
template <typename T, bool TIE_LE, bool DIST_PRE, bool WAVE, bool PUBLISH = false, bool SPEC = false>
__device__ __forceinline__ void cast_body(const RcwDev& p, const uint8_t* __restrict__ actions, const uint8_t* __restrict__ mask,
                                          const int a, const int tid, const int nthr, uint32_t* const lds, const int trace_slot,
                                          uint16_t* __restrict__ spec_out = nullptr, const int spec_cols = 1)
{
    typedef typename Real<T>::vec2 vec2;
    const int H = p.H, HW = p.H * p.W, N = p.N;
    RCW_CAST_STAMP(0);

    // ---- batch 1: the agent's state (all addresses known from the kernel arguments) ------------------------
    // No load sits under a branch (a conditional load is a basic block of its own, and the compiler waits for it at the
    // block's end): absent arrays read a harmless stand-in (the done flag), lanes past the map's end re-read its last word.
    // The three bytes, heading and pose: load_cast_state.
    uint32_t* const tm_hbm = p.tile_map + (size_t)a * p.nwords;
    uint32_t tw[kCastTiles];
#pragma unroll
    for (int k = 0; k < kCastTiles; ++k) {                                  // (vector loads: issued here, awaited below)
        const int t = tid + k * nthr;
        tw[k] = load_at(tm_hbm, (uint32_t)((t < HW ? t : HW - 1) >> 4) * 4u);
    }
    const uint8_t* const mask_q = mask != nullptr ? mask + a : p.done + a;
    const uint8_t* const act_q = actions != nullptr ? actions + a : p.done + a;
    vec2 pos;
    const CastState st = load_cast_state(mask_q, act_q, p.done + a, p.dir + a, Real<T>::pos(p) + a, pos);
    const int m = byte_of_word(st.mask_w, mask_q), was_done = byte_of_word(st.done_w, p.done + a), d = st.d;
    int act = byte_of_word(st.act_w, act_q);
    // the tile words too (and every wavefront of the workgroup has READ the state before lane 0 overwrites it below)
    asm volatile("" :: "v"(tw[0]), "v"(tw[1]), "v"(tw[2]), "v"(tw[3]));
    RCW_CAST_STAMP(1);
    if (mask != nullptr && m == 0) return;
    if (actions == nullptr) act = 0;

    const bool invalid = actions != nullptr && (act < 1 || act > RCW_NUM_ACTIONS);   // @assert SR:140
    if (invalid) act = 0;                                                   // this agent is not stepped
    const bool resample = act != 0 && p.auto_reset != 0 && was_done != 0;
    int d_new = d;
    if (!resample && act == 3) d_new = d + 1 >= p.nd ? 0 : d + 1;          // turn_left  UT:13
    if (!resample && act == 4) d_new = d - 1 < 0 ? p.nd - 1 : d - 1;       // turn_right UT:14

    // ---- batch 2: what depends on the heading ---------------------------------------------------------------
    // (with a re-sampled heading — rare — the row of the OLD heading is fetched for nothing and the right one again below)
    const vec2 dv = Real<T>::dir_table(p)[d];                               // SR:153
    vec2 dvn = dv;                                                          // (SPEC: the heading AFTER the action — the successors move along it)
    if (SPEC) dvn = Real<T>::dir_table(p)[d_new];
    T r_dx[kCastCols], r_dy[kCastCols], r_ddx[kCastCols], r_ddy[kCastCols], r_dot[kCastCols];
    {
        const T* tab = Real<T>::ray_table(p) + (size_t)d_new * RCW_TABLE_ROWS * N;
#pragma unroll
        for (int k = 0; k < kCastCols; ++k) {                               // (five uniform row bases, one lane offset per column)
            const int i = tid + k * nthr;
            const uint32_t o = (uint32_t)(i < N ? i : N - 1) * (uint32_t)sizeof(T);   // (lanes past the last column re-read it)
            r_dx[k] = load_at(tab, o); r_dy[k] = load_at(tab + N, o);
            if (!SPEC) {                                                    // (SPEC derives the other three: spec_derive)
                r_ddx[k] = load_at(tab + 2 * N, o);
                r_ddy[k] = load_at(tab + 3 * N, o); r_dot[k] = load_at(tab + 4 * N, o);
            }
        }
    }

    // LDS: [H guard bytes | H*W tile bytes, the agent's tile map | H guard bytes] (cast_ray_guarded) | the re-sampled pose
    uint8_t* const tb = reinterpret_cast<uint8_t*>(lds) + H;
    // (the loops below run zero times at every BASELINE configuration; `nounroll` keeps the compiler from computing their
    // trip counts — an integer division by blockDim, two dozen instructions each — and from unrolling them by eight)
    if (tid < 2 * H) (tid < H ? tb - H + tid : tb + HW + (tid - H))[0] = 1;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int k = tid + nthr; k < 2 * H; k += nthr) (k < H ? tb - H + k : tb + HW + (k - H))[0] = 1;
    T* const s_pose = reinterpret_cast<T*>(lds + ((HW + 2 * H + 15) / 16) * 4);       // [2] + the heading (auto-reset)
    int& s_pose_d = *reinterpret_cast<int*>(s_pose + 2);
#pragma unroll
    for (int k = 0; k < kCastTiles; ++k) {                                  // (as stage_tile_bytes: the last tile reads as an obstacle)
        const int t = tid + k * nthr;
        if (t < HW) { const uint32_t b = (tw[k] >> ((t & 15) * 2)) & 3u; tb[t] = (uint8_t)(t == HW - 1 ? (b | 1u) : b); }
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int t = tid + kCastTiles * nthr; t < HW; t += nthr) {              // maps of more than kCastTiles * blockDim tiles
        const uint32_t b = (tm_hbm[t >> 4] >> ((t & 15) * 2)) & 3u;
        tb[t] = (uint8_t)(t == HW - 1 ? (b | 1u) : b);
    }
    agent_sync<WAVE>();
    RCW_CAST_STAMP(2);

    // ---- phase 0: dynamics, computed redundantly by every lane (no broadcast needed) --------
    T x = pos.x, y = pos.y;
    int done_now = was_done;                                                // world.done once this call's dynamics are through (SPEC)
    if (resample) {                                                         // wave-uniform, rare
        done_now = 0;
        if (tid == 0) {
            const Pose<T> np = reset_agent<T>(p, a, tm_hbm, nullptr);
            s_pose[0] = np.x; s_pose[1] = np.y; s_pose_d = np.d;
        }
        agent_sync<WAVE>();
        stage_tile_bytes(tb, tm_hbm, HW, tid, nthr);                        // the goal moved
        agent_sync<WAVE>();
        x = s_pose[0]; y = s_pose[1];
        d_new = __builtin_amdgcn_readfirstlane(s_pose_d);
        if (SPEC) dvn = Real<T>::dir_table(p)[d_new];
        const T* tab = Real<T>::ray_table(p) + (size_t)d_new * RCW_TABLE_ROWS * N;
#pragma unroll
        for (int k = 0; k < kCastCols; ++k) {
            const int i = tid + k * nthr;
            const uint32_t o = (uint32_t)(i < N ? i : N - 1) * (uint32_t)sizeof(T);
            r_dx[k] = load_at(tab, o); r_dy[k] = load_at(tab + N, o);
            if (!SPEC) {
                r_ddx[k] = load_at(tab + 2 * N, o);
                r_ddy[k] = load_at(tab + 3 * N, o); r_dot[k] = load_at(tab + 4 * N, o);
            }
        }
    } else if (act != 0) {
        int done = 0;                                                       // reward = done ? goal_reward : zero(R)
        bool oob = false;
        if (act <= 2) {                                                     // SR:150
            const T ix = Real<T>::inc(p) * dv.x, iy = Real<T>::inc(p) * dv.y;
            const T nx = act == 1 ? pos.x + ix : pos.x - ix;                // UT:16-17
            const T ny = act == 1 ? pos.y + iy : pos.y - iy;
            const Collide c = player_colliding<T>(tb, p.H, p.W, nx, ny, Real<T>::radius_sq(p), p.oob_empty);   // SR:162-163
            if (c.wall == 2 || c.goal == 2) oob = true;                     // BoundsError: no mutation
            else if (c.goal) { done = 1; }                                  // SR:166-168
            else if (c.wall) { }                                            // SR:170-171
            else { x = nx; y = ny; }                                        // SR:174
        }
        if (!oob) done_now = done;
        if (tid == 0) {
            if (oob) {
                p.err[0] = RCW_ERR_OUT_OF_BOUNDS;
                p.status[a] = RCW_ERR_OUT_OF_BOUNDS;
            } else {
                Real<T>::pos(p)[a] = Real<T>::make(x, y);                   // SR:174
                p.dir[a] = d_new;                                           // SR:185
                store_reward(p, a, done != 0); p.done[a] = (uint8_t)done;   // SR:167-176, SR:186-187
            }
        }
    }
    if (invalid && tid == 0) { p.err[0] = RCW_ERR_INVALID_ACTION; p.status[a] = RCW_ERR_INVALID_ACTION; }

    // ---- phase 1: one lane per view column (SR:220, SR:401) --------------------------------------------------
    RCW_CAST_STAMP(3);
#ifdef RCW_TRACE_WAVES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // (measurement build: when is the table row here?)
    RCW_CAST_STAMP(4);
#endif
    int32_t* const col_h_a = p.col_h + (size_t)a * N;
    uint8_t* const col_c_a = p.col_c + (size_t)a * N;
    uint32_t* const hc_a = PUBLISH ? p.step_hc + (size_t)a * N : nullptr;
    bool left_the_map = false;
    if (SPEC) {
        // The one-launch step (rcw_fill256_cast_kernel): besides the frame of the state just committed — descriptors as below, and the
        // fill's packed word in slot 0 — the frames of its FOUR SUCCESSORS, one per action of the next act!(world, a) SR:139-191, so
        // that the next launch's fill workgroups only pick the slot the action names.  A move that would be blocked, reach the goal or
        // raise (SR:162-176: the pose stays) has the current frame: its slot gets the current fan's words.  An agent that is done
        // under auto_reset is re-sampled by ANY next action: reset_preview draws the pose the next launch's commit will draw.
        const T* const tab = Real<T>::ray_table(p) + (size_t)d_new * RCW_TABLE_ROWS * N;
        uint16_t* const slot_a = spec_out + (size_t)a * 5u * (size_t)N;     // [B][5][N]
        const uint32_t stride = (uint32_t)N;
        int32_t* const ch = spec_cols ? col_h_a : nullptr;                  // the descriptors of the current frame: only where somebody reads them (rcw_api.hip, ensure_columns)
        spec_derive_rows<T>(dvn.x, dvn.y, r_dx, r_dy, r_ddx, r_ddy, r_dot);
        const bool reborn = p.auto_reset != 0 && done_now != 0;
        bool f_free = false, b_free = false;
        const T ix = Real<T>::inc(p) * dvn.x, iy = Real<T>::inc(p) * dvn.y;
        const T xf = x + ix, yf = y + iy, xb = x - ix, yb = y - iy;        // UT:16-17
        if (!reborn) {
            const Collide cf = player_colliding<T>(tb, p.H, p.W, xf, yf, Real<T>::radius_sq(p), p.oob_empty);
            const Collide cb = player_colliding<T>(tb, p.H, p.W, xb, yb, Real<T>::radius_sq(p), p.oob_empty);
            f_free = cf.wall == 0 && cf.goal == 0;
            b_free = cb.wall == 0 && cb.goal == 0;
        }
        const uint32_t stay = 1u | (!reborn && !f_free ? 2u : 0u) | (!reborn && !b_free ? 4u : 0u);
        left_the_map = spec_fan<T, TIE_LE, DIST_PRE, true>(p, tb, tid, nthr, x, y, r_dx, r_dy, r_ddx, r_ddy, r_dot, tab, dvn.x, dvn.y, ch, col_c_a, slot_a, stride, stay);
#ifdef RCW_DEV_SWITCHES
        if (p.spec_debug & 16) return;                                      // (timing probe: the current state's fan only)
#endif
        if (!reborn) {
            if (f_free) (void)spec_fan<T, TIE_LE, DIST_PRE, false>(p, tb, tid, nthr, xf, yf, r_dx, r_dy, r_ddx, r_ddy, r_dot, tab, dvn.x, dvn.y, nullptr, nullptr, slot_a, stride, 2u);
            if (b_free) (void)spec_fan<T, TIE_LE, DIST_PRE, false>(p, tb, tid, nthr, xb, yb, r_dx, r_dy, r_ddx, r_ddy, r_dot, tab, dvn.x, dvn.y, nullptr, nullptr, slot_a, stride, 4u);
#pragma unroll
            for (int turn = 0; turn < 2; ++turn) {
                const int dt = turn == 0 ? (d_new + 1 >= p.nd ? 0 : d_new + 1) : (d_new - 1 < 0 ? p.nd - 1 : d_new - 1);   // UT:13-14
                const T* const tt = Real<T>::ray_table(p) + (size_t)dt * RCW_TABLE_ROWS * N;
                const vec2 dvt = Real<T>::dir_table(p)[dt];
#ifdef RCW_DEV_SWITCHES
                if (!(p.spec_debug & 8))                                    // (timing probe: the turns with the current heading's rows, no further table loads)
#endif
                spec_load_rows<T>(tt, tid, nthr, N, r_dx, r_dy);
                spec_derive_rows<T>(dvt.x, dvt.y, r_dx, r_dy, r_ddx, r_ddy, r_dot);
                (void)spec_fan<T, TIE_LE, DIST_PRE, false>(p, tb, tid, nthr, x, y, r_dx, r_dy, r_ddx, r_ddy, r_dot, tt, dvt.x, dvt.y, nullptr, nullptr, slot_a, stride, turn == 0 ? 8u : 16u);
            }
        } else {
            agent_sync<WAVE>();                                             // (every lane has read the tile bytes of the done state)
            if (tid == 0) {
                const Pose<T> np = reset_preview<T>(p, a, tb);
                s_pose[0] = np.x; s_pose[1] = np.y; s_pose_d = np.d;
            }
            agent_sync<WAVE>();
            const T xr = s_pose[0], yr = s_pose[1];
            const int dr = __builtin_amdgcn_readfirstlane(s_pose_d);
            const T* const tt = Real<T>::ray_table(p) + (size_t)dr * RCW_TABLE_ROWS * N;
            const vec2 dvr = Real<T>::dir_table(p)[dr];
            spec_load_rows<T>(tt, tid, nthr, N, r_dx, r_dy);
            spec_derive_rows<T>(dvr.x, dvr.y, r_dx, r_dy, r_ddx, r_ddy, r_dot);
            (void)spec_fan<T, TIE_LE, DIST_PRE, false>(p, tb, tid, nthr, xr, yr, r_dx, r_dy, r_ddx, r_ddy, r_dot, tt, dvr.x, dvr.y, nullptr, nullptr, slot_a, stride, 30u);
        }
        if (left_the_map) { p.err[0] = RCW_ERR_OUT_OF_BOUNDS; p.status[a] = RCW_ERR_OUT_OF_BOUNDS; }   // (Julia: BoundsError in cast_ray)
        return;
    }
#pragma unroll
    for (int k = 0; k < kCastCols; ++k) {
        const int i = tid + k * nthr;
        if (i < N) left_the_map |= cast_column<T, TIE_LE, DIST_PRE, PUBLISH>(p, tb, col_h_a, col_c_a, i, x, y, r_dx[k], r_dy[k], r_ddx[k], r_ddy[k], r_dot[k], hc_a);
#ifdef RCW_TRACE_WAVES
        if (k == 0) RCW_CAST_STAMP(5);
#endif
    }
    RCW_CAST_STAMP(6);
#ifdef RCW_TRACE_WAVES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // (the descriptor stores acknowledged)
    RCW_CAST_STAMP(7);
    if (tid == 0 && trace_slot < 4096) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        g_cast_trace[trace_slot * 10 + 8] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
#endif
    if (N > kCastCols * nthr) {                                             // more than kCastCols columns a lane (N > 1024)
        const T* tab = Real<T>::ray_table(p) + (size_t)d_new * RCW_TABLE_ROWS * N;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int i = tid + kCastCols * nthr; i < N; i += nthr)
            left_the_map |= cast_column<T, TIE_LE, DIST_PRE, PUBLISH>(p, tb, col_h_a, col_c_a, i, x, y, tab[i], tab[N + i], tab[2 * N + i], tab[3 * N + i], tab[4 * N + i], hc_a);
    }
    if (left_the_map) { p.err[0] = RCW_ERR_OUT_OF_BOUNDS; p.status[a] = RCW_ERR_OUT_OF_BOUNDS; }   // (Julia: BoundsError in cast_ray)
}

template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_cast_kernel(const RcwDev p,
                                                          const uint8_t* __restrict__ actions,
                                                          const uint8_t* __restrict__ mask, int first)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // agents [first, first + gridDim.x): one workgroup each
    cast_body<T, TIE_LE, DIST_PRE, false>(p, actions, mask, first + (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, lds, (int)blockIdx.x);
}

#ifdef RCW_DEV_SWITCHES
// Development build only (RCW_CAST_WAVES=1), measured and rejected (docs/experiments.md): the same with a WAVEFRONT per agent, four agents a
// workgroup — a quarter of the workgroups for the dispatcher to hand out, no s_barrier.  The 4096 workgroups of cfg-2 then enter within
// 0.9 us instead of 1.6, and the kernel takes 11.5 us instead of 11.2 (cfg-4's shard: 20.2 vs 19.3): the wavefronts live longer by what
// they no longer wait to be dispatched — the SIMDs' issue slots bound the kernel, not the dispatcher.  agents [first, last).
template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_cast_waves_kernel(const RcwDev p,
                                                                const uint8_t* __restrict__ actions,
                                                                const uint8_t* __restrict__ mask, int first, int last, int lds_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int a = first + (int)blockIdx.x * (kBlock / 64) + wave;
    if (a >= last) return;                                                  // (wave-uniform: the batch's last workgroup may be short)
    cast_body<T, TIE_LE, DIST_PRE, true>(p, actions, mask, a, (int)(threadIdx.x & 63u), 64, lds + (size_t)wave * lds_words, a - first);
}
#endif   // RCW_DEV_SWITCHES (rcw_cast_waves_kernel)

// ---- kernel 2 of a step: column descriptors -> pixels -------------------------------------
// The bandwidth kernel: B·N·H_cam·4 bytes, written once.  HBM on MI355X takes writes fastest
// when the whole chip sweeps ONE compact window through memory (measured, tools/fill_bench:
// 6.6–6.7 TB/s for 128–256 workgroups striding a 0.5–1 MiB window, against 5.9–6.2 TB/s
// for one workgroup per 256 KiB frame and 4.5 TB/s for 1024 workgroups).  So the grid is
// small and fixed (p.fill_grid workgroups of 4 wavefronts); wavefront g of G writes the 1 KiB
// chunks g, g+G, g+2G, ... of the flat (H_cam, N, B) batch.  At H_cam = 256 a chunk is one
// image column: the wavefront prefetches 64 descriptors (one per lane, for its next 64
// chunks), then per chunk broadcasts one with v_readlane and lane l writes rows 4l..4l+3
// with one 16-byte store — 64 lanes x 16 B = the whole column in one instruction.
#ifdef RCW_TRACE_WAVES
// Measurement build only (make trace -> lib/librcw_hip_trace.so, tools/wave_trace.py): rcw_fill256_kernel's wavefronts
// leave the time (s_memrealtime, 100 MHz, one clock for the whole device) at which each of their groups starts its
// descriptor loads and has them back, the time they end, and where they ran (HW_ID, XCC_ID).
__device__ unsigned long long g_wave_trace[1024 * 40];
}  // namespace
extern "C" __attribute__((visibility("default"))) int rcw_wave_trace_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 1024 * 40);
}
namespace {
#endif
template <bool PLAIN>
__device__ __forceinline__ void store16(u32x4* dst, u32x4 v)
{
    if (PLAIN) *dst = v; else __builtin_nontemporal_store(v, dst);
}

// rcw_fill256_kernel's body as a function — workgroup `block` of `blocks` — for rcw_fill256_draw_kernel, which runs it in the
// first `blocks` workgroups of a larger launch (the kernel proper follows, with its body verbatim)
template <bool PLAIN, int EXTRA = 0>
__device__ __forceinline__ void fill256_body(const RcwDev& p, const int32_t* __restrict__ col_h, const uint8_t* __restrict__ col_c,
                                             u32x4* __restrict__ out, long long total_cols, const uint8_t* __restrict__ mask,
                                             int block, int blocks)
{
    const int lane = threadIdx.x & 63;
    const long long G = (long long)blocks * (kBlock / 64);
    const long long g = (long long)block * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    const int r0 = lane * 4;
    for (long long base = g; base < total_cols; base += G * 64) {
        // lane l holds the descriptor of this wavefront's l-th next chunk
        const long long mine = base + (long long)lane * G;
        int pad_l = -1;                       // -1: nothing to write (past the end / masked out)
        uint32_t colour_l = 0u;
        if (mine < total_cols && (mask == nullptr || mask[mine / p.N] != 0)) {
            // THREE DEPENDENT round trips, on purpose: height -> colour id -> colour.  This prefetch is part of the kernel's pace
            // (docs/experiments.md): every shorter form measured — the two loads issued together, a packed word, the colour by
            // selects — makes the kernel SLOWER, and the more so the larger the batch.  Round 4 found that out a fourth time: with
            // this body moved into a function the compiler issued both loads at once, and the fill of an 8 GiB batch took 1420 us
            // instead of 1250 (1 GiB: 158 instead of 156.5).  The empty asm statements pin the order the round-1 kernel had.
            int h = col_h[mine];
            asm volatile("" :: "v"(h) : "memory");
#ifdef RCW_DEV_SWITCHES
            // (development experiment RCW_FILL_TRIPS: EXTRA more dependent round trips in front — the same word again, at an
            // address the compiler cannot tell from the first one's)
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) {
                int z = h & 0x40000000;                                     // 0: a height is below 2^30
                asm volatile("" : "+v"(z));
                h = col_h[mine + z];
                asm volatile("" :: "v"(h) : "memory");
            }
#endif
            pad_l = column_padding(256, h);
            const uint32_t cid = col_c[mine];
            asm volatile("" :: "v"(cid) : "memory");
            colour_l = p.colour[cid & 3];
        }
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            const int pad = __builtin_amdgcn_readlane(pad_l, l);
            if (pad < 0) continue;            // wave-uniform
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)colour_l, l);
            u32x4 v;
            v.x = pixel(r0 + 0, pad, 256, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, 256, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, 256, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, 256, c, ceil_c, floor_c);
            store16<PLAIN>(out + (base + (long long)l * G) * 64 + lane, v);
        }
    }
}

// (the kernel proper, with the round-1 body verbatim rather than through fill256_body: its generated code — in particular the
// prefetch's three dependent round trips — is what every measurement of rounds 1-3 was made with; tests/test_build_checks.py
// checks that shape on the ISA of both)
template <bool PLAIN>
__global__ __launch_bounds__(kBlock) void rcw_fill256_kernel(const RcwDev p,
                                                             const int32_t* __restrict__ col_h,
                                                             const uint8_t* __restrict__ col_c,
                                                             u32x4* __restrict__ out, long long total_cols,
                                                             const uint8_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const long long G = (long long)gridDim.x * (kBlock / 64);
    const long long g = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    const int r0 = lane * 4;
#ifdef RCW_TRACE_WAVES
    int grp = 0;
#endif
    for (long long base = g; base < total_cols; base += G * 64) {
#ifdef RCW_TRACE_WAVES
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
        // lane l holds the descriptor of this wavefront's l-th next chunk
        const long long mine = base + (long long)lane * G;
        int pad_l = -1;                       // -1: nothing to write (past the end / masked out)
        uint32_t colour_l = 0u;
        if (mine < total_cols && (mask == nullptr || mask[mine / p.N] != 0)) {
            pad_l = column_padding(256, col_h[mine]);
            colour_l = p.colour[col_c[mine] & 3];
        }
#ifdef RCW_TRACE_WAVES
        asm volatile("" : "+v"(pad_l), "+v"(colour_l));
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && g < 1024 && grp < 18) { g_wave_trace[(g * 20 + grp) * 2] = t0; g_wave_trace[(g * 20 + grp) * 2 + 1] = t1; }
        grp += 1;
#endif
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            const int pad = __builtin_amdgcn_readlane(pad_l, l);
            if (pad < 0) continue;            // wave-uniform
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)colour_l, l);
            u32x4 v;
            v.x = pixel(r0 + 0, pad, 256, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, 256, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, 256, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, 256, c, ceil_c, floor_c);
            store16<PLAIN>(out + (base + (long long)l * G) * 64 + lane, v);
        }
    }
#ifdef RCW_TRACE_WAVES
    if (lane == 0 && g < 1024) {
        g_wave_trace[(g * 20 + 19) * 2] = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        g_wave_trace[(g * 20 + 18) * 2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
#endif
}

#ifdef RCW_DEV_SWITCHES
// Development build only (RCW_FILL_TRIPS=1..3): rcw_fill256_kernel's body with that many MORE dependent round trips in each group's
// prefetch (five measurements say a shorter prefetch makes this store stream slower: is a longer one faster?)
template <int EXTRA>
__global__ __launch_bounds__(kBlock) void rcw_fill256_trips_kernel(const RcwDev p, const int32_t* __restrict__ col_h, const uint8_t* __restrict__ col_c,
                                                                   u32x4* __restrict__ out, long long total_cols, const uint8_t* __restrict__ mask)
{
    fill256_body<false, EXTRA>(p, col_h, col_c, out, total_cols, mask, (int)blockIdx.x, (int)gridDim.x);
}
#endif
#ifdef RCW_DEV_SWITCHES
// Development build only (RCW_STEP_FUSED=1): the WHOLE step in one launch.  Workgroups 0 .. fill_blocks - 1 are the camera fill's
// (dispatched first, onto an empty device, one per CU as in a launch of their own), the others cast four agents each, a wavefront per
// agent (rcw_cast_waves_kernel).  The hand-off inside the launch (MI355X_MICROARCH.md, "inter-workgroup visibility"): a casting
// wavefront stores each column's {height, colour id} as ONE word with an agent-scope (write-through, sc1) store into p.step_hc, waits
// for all its stores (s_waitcnt vmcnt(0)), then its first lane stores the step's epoch into p.step_flags[agent], sc1 too; a fill
// wavefront's lane polls the flag of the agent its next chunk belongs to with sc1 loads (they bypass the CU's L1, which still holds the
// previous step's lines) and reads the word with an sc1 load only after the flag has matched.  Casting workgroups never wait, so
// they drain whatever the dispatcher's order; a fill lane gives up after ~1 s (RCW_ERR_HIP in the handle's error word).
template <bool PLAIN>
__device__ __forceinline__ void fill256_wait_body(const RcwDev& p, u32x4* __restrict__ out, long long total_cols, const uint8_t* __restrict__ mask,
                                                  int block, int blocks, uint32_t epoch)
{
    const int lane = threadIdx.x & 63;
    const long long G = (long long)blocks * (kBlock / 64);
    const long long g = (long long)block * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    const int r0 = lane * 4;
    for (long long base = g; base < total_cols; base += G * 64) {
        const long long mine = base + (long long)lane * G;
        int pad_l = -1;
        uint32_t colour_l = 0u;
        if (mine < total_cols) {
            const uint32_t a = (uint32_t)mine / (uint32_t)p.N;              // (the launcher takes this form below 2^31 columns only)
            if (mask == nullptr || mask[a] != 0) {
                int spins = 0;
                uint32_t hc;
                if (p.step_fused == 2) {
                    // the word carries the step's epoch (its low 21 bits) as a tag: no flag, one round trip when the word is there
                    while (((hc = __hip_atomic_load(p.step_hc + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 11) != (epoch & 0x1fffffu)) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1 << 20)) { p.err[0] = RCW_ERR_HIP; break; }
                    }
                } else {
                    const uint32_t* const flag = p.step_flags + a;
                    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1 << 20)) { p.err[0] = RCW_ERR_HIP; break; }
                    }
                    asm volatile("" ::: "memory");
                    hc = __hip_atomic_load(p.step_hc + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("" :: "v"(hc) : "memory");
                pad_l = (int)(hc & 0x1ffu);
                colour_l = p.colour[(hc >> 9) & 3u];
            }
        }
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            const int pad = __builtin_amdgcn_readlane(pad_l, l);
            if (pad < 0) continue;            // wave-uniform
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)colour_l, l);
            u32x4 v;
            v.x = pixel(r0 + 0, pad, 256, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, 256, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, 256, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, 256, c, ceil_c, floor_c);
            store16<PLAIN>(out + (base + (long long)l * G) * 64 + lane, v);
        }
    }
}

template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_step256_kernel(const RcwDev p, const uint8_t* __restrict__ actions, const uint8_t* __restrict__ mask,
                                                             u32x4* __restrict__ out, long long total_cols, int fill_blocks, int lds_words,
                                                             uint32_t epoch)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if ((int)blockIdx.x < fill_blocks) { fill256_wait_body<false>(p, out, total_cols, mask, (int)blockIdx.x, fill_blocks, epoch); return; }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int a = ((int)blockIdx.x - fill_blocks) * (kBlock / 64) + wave;
    if (a >= p.B) return;                                                   // (wave-uniform: the batch's last workgroup may be short)
    cast_body<T, TIE_LE, DIST_PRE, true, true>(p, actions, mask, a, (int)(threadIdx.x & 63u), 64, lds + (size_t)wave * lds_words, a);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // every store of this wavefront has been acknowledged
    if ((threadIdx.x & 63u) == 0) __hip_atomic_store(p.step_flags + a, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif   // RCW_DEV_SWITCHES (rcw_step256_kernel)

// ---- the WHOLE step in one launch, without a dependency inside it (round 6) ------------------------------------------------
// act!(env, a) SR:333-340 orders dynamics -> cast_rays! -> update_camera_view!; as two kernels the cast (11 us at 4096 agents x
// 256 columns, latency / issue bound) and a launch boundary sit in front of every fill.  But the frame of step t + 1 depends only on
// (state_t, action_{t+1}) and there are four actions: the casting workgroups of launch t, once they have committed act!(world, a_t),
// also cast the four successors of the new state into five slots of packed column words [B][5][N] (slot 0: the state itself — an
// invalid action leaves the agent where it is; slots 1..4: the actions), and the fill workgroups of launch t + 1 only read the action
// and pick the slot: action -> word -> colour, three dependent round trips a group like rcw_fill256_kernel's height -> colour id ->
// colour (its pace: DESIGN.md §4.2).  Nothing in a launch waits for anything else in it — unlike rcw_step256_kernel above, whose
// fill workgroups waited for the cast's flags and gained nothing.  Two slot buffers alternate: launch t reads the one launch t - 1
// wrote and writes the other.  Workgroups 0 .. fill_blocks - 1 are the fill's (dispatched first, one per CU as in a launch of their
// own); the casting workgroups — VALU / LDS work — run beside them under the HBM-bound sweep.  rcw_cast_successors_kernel is the
// casting half alone: it PRIMES the slots behind a reset / set_state (or a first step), the camera fill following as a launch of its own.
template <bool PLAIN>
__device__ __forceinline__ void fill256_spec_body(const RcwDev& p, const uint8_t* __restrict__ actions, const uint16_t* __restrict__ slots,
                                                  u32x4* __restrict__ out, long long total_cols, int block, int blocks, int n_shift)
{
    // (Raising the fill wavefronts' priority over the casting ones — s_setprio 3 — changes nothing: what the casting half costs this
    // launch is its memory operations, not its issue slots: profiles/r06_step_forms.txt.)
    const int lane = threadIdx.x & 63;
    const long long G = (long long)blocks * (kBlock / 64);
    const long long g = (long long)block * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    const int r0 = lane * 4;
    for (long long base = g; base < total_cols; base += G * 64) {
        // lane l holds the word of this wavefront's l-th next chunk
        const long long mine = base + (long long)lane * G;
        int pad_l = -1;                       // -1: nothing to write (past the end)
        uint32_t colour_l = 0u;
        if (mine < total_cols) {
            // the chunk's agent (the launcher takes this form below 2^29 columns), its action, the slot the action names
            const uint32_t a = n_shift >= 0 ? (uint32_t)mine >> n_shift : (uint32_t)mine / (uint32_t)p.N;   // (n_shift: log2(N) where N is a power of two, else -1)
            const uint32_t act = actions[a];
            asm volatile("" :: "v"(act) : "memory");
            const uint32_t sel = act - 1u < (uint32_t)RCW_NUM_ACTIONS ? act : 0u;   // (an action outside 1..4: the agent is not stepped, SR:140)
            const uint32_t w = slots[(size_t)mine + (size_t)(4u * a + sel) * (size_t)p.N];   // [B][5][N]: (5 a + sel) N + (mine - a N)
            asm volatile("" :: "v"(w) : "memory");
            pad_l = (int)(w & 0x1ffu);
            colour_l = p.colour[(w >> 9) & 3u];
        }
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            const int pad = __builtin_amdgcn_readlane(pad_l, l);
            if (pad < 0) continue;            // wave-uniform
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)colour_l, l);
            u32x4 v;
            v.x = pixel(r0 + 0, pad, 256, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, 256, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, 256, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, 256, c, ceil_c, floor_c);
            store16<PLAIN>(out + (base + (long long)l * G) * 64 + lane, v);
        }
    }
}

// casting workgroup `block` of the launch.  WAVE: a wavefront per agent, four agents a casting workgroup (at most 256 view columns:
// four a lane); else a workgroup per agent.
template <typename T, bool TIE_LE, bool DIST_PRE, bool WAVE>
__device__ __forceinline__ void cast_successors(const RcwDev& p, const uint8_t* __restrict__ actions, const uint8_t* __restrict__ mask, int block,
                                                uint16_t* __restrict__ slots_out, uint32_t* lds, int lds_words, int cols)
{
    if (WAVE) {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int a = block * (kBlock / 64) + wave;
        if (a >= p.B) return;                                               // (wave-uniform: the batch's last workgroup may be short)
        cast_body<T, TIE_LE, DIST_PRE, true, false, true>(p, actions, mask, a, (int)(threadIdx.x & 63u), 64, lds + (size_t)wave * lds_words, a, slots_out, cols);
    } else {
        cast_body<T, TIE_LE, DIST_PRE, false, false, true>(p, actions, mask, block, (int)threadIdx.x, kBlock, lds, block, slots_out, cols);
    }
}

template <typename T, bool TIE_LE, bool DIST_PRE, bool WAVE>
__global__ __launch_bounds__(kBlock) void rcw_fill256_cast_kernel(const RcwDev p, const uint8_t* __restrict__ actions, const uint8_t* __restrict__ mask,
                                                                  u32x4* __restrict__ out, long long total_cols, int fill_blocks,
                                                                  const uint16_t* __restrict__ slots_in, uint16_t* __restrict__ slots_out, int lds_words, int n_shift, int cols)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
#ifdef RCW_DEV_SWITCHES
    if (p.spec_debug & ((int)blockIdx.x < fill_blocks ? 2 : 1)) return;     // (timing probes: one half of the launch alone)
#endif
    if ((int)blockIdx.x < fill_blocks) { fill256_spec_body<false>(p, actions, slots_in, out, total_cols, (int)blockIdx.x, fill_blocks, n_shift); return; }
    cast_successors<T, TIE_LE, DIST_PRE, WAVE>(p, actions, mask, (int)blockIdx.x - fill_blocks, slots_out, lds, lds_words, cols);
}

// The casting workgroups alone, under a name of their own (profiles tell a step from what primes its slots): behind rcw_reset /
// rcw_set_state — no action, maybe a mask: masked-out agents keep their slots — or for a first step; the camera fill follows as a launch
// of its own, from the descriptors.
template <typename T, bool TIE_LE, bool DIST_PRE, bool WAVE>
__global__ __launch_bounds__(kBlock) void rcw_cast_successors_kernel(const RcwDev p, const uint8_t* __restrict__ actions, const uint8_t* __restrict__ mask,
                                                                     uint16_t* __restrict__ slots_out, int lds_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    cast_successors<T, TIE_LE, DIST_PRE, WAVE>(p, actions, mask, (int)blockIdx.x, slots_out, lds, lds_words, 1);
}

// The same moving window for the camera heights that tile a 1 KiB chunk evenly: H_cam = 256·k (a chunk is one of
// the k row blocks of a column: M = 1) and H_cam = 128 or 64 (a chunk holds M = 2 or 4 whole columns; lane l of a
// chunk belongs to column l / (64 / M)).  Lane l prefetches the descriptor(s) of the wavefront's l-th next chunk, as
// above; with M > 1 every lane then picks its column's out of the M broadcast ones.
template <int M>
__global__ __launch_bounds__(kBlock) void rcw_fill_window_kernel(const RcwDev p,
                                                                 const int32_t* __restrict__ col_h,
                                                                 const uint8_t* __restrict__ col_c,
                                                                 u32x4* __restrict__ out, long long total_chunks,
                                                                 const uint8_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const long long G = (long long)gridDim.x * (kBlock / 64);
    const long long g = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    const int Hc = p.Hc;
    const int k = M == 1 ? Hc >> 8 : 1;                    // chunks per column (M == 1)
    const int sub = M == 1 ? 0 : lane / (64 / M);          // this lane's column within a chunk (M > 1)
    const int r_lane = M == 1 ? lane * 4 : (lane - sub * (64 / M)) * 4;
    for (long long base = g; base < total_chunks; base += G * 64) {
        const long long mine = base + (long long)lane * G;
        int pad_l[M], rb_l = 0;
        uint32_t colour_l[M];
#pragma unroll
        for (int j = 0; j < M; ++j) { pad_l[j] = -1; colour_l[j] = 0u; }
        if (mine < total_chunks) {
            const long long col0 = M == 1 ? mine / k : mine * M;             // first (only) column of the chunk
            if (mask == nullptr || mask[col0 / p.N] != 0) {                   // (a chunk never spans two agents: N*Hc % 256 == 0 here)
                rb_l = M == 1 ? (int)(mine - col0 * k) * 256 : 0;
                if constexpr (M == 1) {
                    pad_l[0] = column_padding(Hc, col_h[col0]);
                    colour_l[0] = p.colour[col_c[col0] & 3];
                } else {
                    // the chunk's M columns start at a multiple of M: ONE M·4-byte and ONE M-byte load instead of 2·M scattered ones
                    // (H_cam 64: 176 -> 163 µs per GiB, 128: 160 -> 158; at M = 4 the colours by selects, not by M more — dependent —
                    // loads from the kernel argument's array)
                    struct __attribute__((aligned(4 * M))) Heights { int32_t h[M]; };
                    struct __attribute__((aligned(M))) Ids { uint8_t c[M]; };
                    const Heights hw = *reinterpret_cast<const Heights*>(col_h + col0);
                    const Ids cw = *reinterpret_cast<const Ids*>(col_c + col0);
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        pad_l[j] = column_padding(Hc, hw.h[j]);
                        if constexpr (M == 4) {
                            const uint32_t id = cw.c[j];
                            const uint32_t lo = (id & 1u) ? p.colour[1] : p.colour[0], hi = (id & 1u) ? p.colour[3] : p.colour[2];
                            colour_l[j] = (id & 2u) ? hi : lo;
                        } else {
                            colour_l[j] = p.colour[cw.c[j] & 3];
                        }
                    }
                }
            }
        }
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            const int pad0 = __builtin_amdgcn_readlane(pad_l[0], l);
            if (pad0 < 0) continue;            // wave-uniform: past the end / masked out
            int pad = pad0;
            uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)colour_l[0], l);
#pragma unroll
            for (int j = 1; j < M; ++j) {
                const int pj = __builtin_amdgcn_readlane(pad_l[j], l);
                const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)colour_l[j], l);
                pad = sub == j ? pj : pad;
                c = sub == j ? cj : c;
            }
            const int r0 = (M == 1 ? __builtin_amdgcn_readlane(rb_l, l) : 0) + r_lane;
            u32x4 v;
            v.x = pixel(r0 + 0, pad, Hc, c, ceil_c, floor_c);
            v.y = pixel(r0 + 1, pad, Hc, c, ceil_c, floor_c);
            v.z = pixel(r0 + 2, pad, Hc, c, ceil_c, floor_c);
            v.w = pixel(r0 + 3, pad, Hc, c, ceil_c, floor_c);
            __builtin_nontemporal_store(v, out + (base + (long long)l * G) * 64 + lane);
        }
    }
}

// The moving window for ANY camera height of at least 24 rows (height_camera_view_pu is a free kwarg, SR:271): a chunk
// is 256 consecutive pixels of the flat (H_cam, N, B) batch, whatever columns they belong to.  A chunk that starts at
// row rem0 of its first column touches at most K = 254 / H_cam + 2 columns; lane l of the prefetch finds (first
// column, rem0) of the wavefront's l-th next chunk — carried from group to group as (quotient, remainder), no division
// — loads those K columns' descriptors and parks them as (padding | valid << 31, colour) pairs in wave-private LDS.  In
// the chunk loop a lane derives its own column from its flat pixel offset rem0 + 4 lane WITHOUT a division: with
// 4 lane = qv·H_cam + rv fixed per lane and rem0 < H_cam, the column is qv + (rem0 + rv >= H_cam) and the row
// rem0 + rv less H_cam in that case — an add, a subtract, an unsigned min, a compare and an add-with-carry.  It then
// reads that column's pair back with one ds_read_b64 and writes its four pixels with one 16-byte store, as the kernels
// above.  ALIGNED (H_cam % 4 == 0): the four pixels never straddle a column; otherwise the lane also reads the next
// column's pair and picks per pixel.  A group whose 64 chunks are all whole and unmasked (nearly every one) runs a
// loop without branches that fetches the next chunk's pair while it computes this one's pixels — one wavefront per
// SIMD has nobody else to hide the LDS round trip behind; chunks at a masked agent's border and the batch's last,
// short chunk take the general loop with its per-pixel path.
struct FlatLane { int qv, rv; };          // 4 lane = qv · height + rv
__device__ __forceinline__ FlatLane flat_lane(int lane, int height)
{
    FlatLane L;
    L.qv = (4 * lane) / height;
    L.rv = 4 * lane - L.qv * height;
    return L;
}
// (column relative to the chunk's first, row in it) of this lane's first pixel, for a chunk that starts at row rem0
__device__ __forceinline__ void flat_locate(const FlatLane& L, int rem0, int height, int& rel, int& r)
{
    const uint32_t t = (uint32_t)(rem0 + L.rv), u = t - (uint32_t)height;
    r = (int)(t < u ? t : u);                                              // v_min_u32: u wraps when t < height
    rel = L.qv + (t >= (uint32_t)height ? 1 : 0);
}

// a column's descriptor in LDS: x = rows of ceiling (SR:436), y = H_cam - x (the first row of floor), z = colour, w = valid.
// The lane's first pixel is row r of column d0 (the following pixels may lie in d1 when H_cam % 4 != 0); pixel e is
// ceiling while e < x - r, colour while e < y - r, else floor (SR:437-439): two subtractions, then compares with constants.
template <bool ALIGNED>
__device__ __forceinline__ u32x4 flat_fill_pixels(int r, int Hc, uint4 d0, uint4 d1, uint32_t ceil_c, uint32_t floor_c, bool* ok)
{
    u32x4 v;
    const int a0 = (int)d0.x - r, b0 = (int)d0.y - r;
    if (ALIGNED) {
        v.x = 0 < a0 ? ceil_c : (0 < b0 ? d0.z : floor_c);
        v.y = 1 < a0 ? ceil_c : (1 < b0 ? d0.z : floor_c);
        v.z = 2 < a0 ? ceil_c : (2 < b0 ? d0.z : floor_c);
        v.w = 3 < a0 ? ceil_c : (3 < b0 ? d0.z : floor_c);
        ok[0] = ok[1] = ok[2] = ok[3] = d0.w != 0u;
    } else {
        const int c = Hc - r;                                              // pixels e >= c are in the following column, from its row 0
        const int a1 = (int)d1.x + c, b1 = (int)d1.y + c;                   // (row e - c of d1: e - c < x  <=>  e < x + c)
        uint32_t px[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool next = e >= c;
            const int a = next ? a1 : a0, b = next ? b1 : b0;
            px[e] = e < a ? ceil_c : (e < b ? (next ? d1.z : d0.z) : floor_c);
            ok[e] = (next ? d1.w : d0.w) != 0u;
        }
        v.x = px[0]; v.y = px[1]; v.z = px[2]; v.w = px[3];
    }
    return v;
}

// PAIR (development build, RCW_FILL_FLAT_PAIRS=1: measured, profiles/r05_flat_kernels.txt): workgroups of EIGHT wavefronts, two to a
// slot of the window — wavefronts w and w + 4 make the same group's descriptors and take its chunks by turns (t even / odd) —: the
// same compact window, twice the issue slots (a wavefront alone on its SIMD issues a vector instruction every four cycles, two
// wavefronts one every two).
template <bool ALIGNED, int K, bool PAIR = false>                       // K = the columns a chunk may touch (254 / H_cam + 2)
__global__ __launch_bounds__(PAIR ? 2 * kBlock : kBlock) void rcw_fill_flat_kernel(const RcwDev p,
                                                               const int32_t* __restrict__ col_h,
                                                               const uint8_t* __restrict__ col_c,
                                                               uint32_t* __restrict__ out, long long total_cols,
                                                               const uint8_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63;
    constexpr int STEP = PAIR ? 2 : 1;                                    // chunks of a group between two of this wavefront's
    const int half = PAIR ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;   // which of a slot's two wavefronts
    const uint32_t G = gridDim.x * (kBlock / 64);
    const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3u);
    uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    asm volatile("" : "+v"(ceil_c), "+v"(floor_c));                         // (in vector registers once: a v_cndmask reads at most one scalar)
    const int Hc = p.Hc;
    constexpr int KS = K + 1;                                             // (one spare pair per chunk: the straddling read of the last column)
    const unsigned long long total_px = (unsigned long long)total_cols * (unsigned)Hc;
    const unsigned long long total_chunks = (total_px + 255) >> 8;
    uint4* const desc = reinterpret_cast<uint4*>(lds) + (size_t)(threadIdx.x >> 6) * 64 * KS;   // [64 chunks][KS]
    const FlatLane L = flat_lane(lane, Hc);
    // this lane's chunk of the first group, as (first column, row in it); every group moves all lanes by the same pixels
    const unsigned long long id0 = (unsigned long long)g + (unsigned long long)lane * G;
    const unsigned long long step_px = (unsigned long long)G * 64 * 256;
    const uint32_t dq = (uint32_t)(step_px / (unsigned)Hc), dr = (uint32_t)(step_px - (unsigned long long)dq * (unsigned)Hc);
    uint32_t col = (uint32_t)((id0 * 256) / (unsigned)Hc);
    uint32_t rem = (uint32_t)(id0 * 256 - (unsigned long long)col * (unsigned)Hc);
    u32x4* const out4 = reinterpret_cast<u32x4*>(out);
    const unsigned long long dstep = (unsigned long long)G * 64;
    const uint32_t last_col = (uint32_t)total_cols - 1u;
#ifdef RCW_TRACE_WAVES
    int grp = 0;
#endif
    for (unsigned long long base = g; base < total_chunks; base += (unsigned long long)G * 64) {
#ifdef RCW_TRACE_WAVES
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
        const unsigned long long id = base + (unsigned long long)lane * G;
        const bool exists = id < total_chunks;
        int touched = 0;                                                     // last column of the chunk, relative
#pragma unroll
        for (int k = 1; k < K; ++k) touched += (rem + 255u >= (unsigned)(k * Hc)) ? 1 : 0;
        if (!exists) touched = -1;
        bool all_valid = exists && (id + 1) * 256 <= total_px;
        // (all loads first — addresses clamped into the arrays, not predicated —, then everything that uses one)
        int32_t hh[KS];
        uint32_t cc[KS], mm[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const uint32_t c = min(col + (unsigned)j, last_col);
#ifdef RCW_DEV_SWITCHES
            if (p.fill_pairs == 2) { hh[j] = (int32_t)(c & 127u); cc[j] = c & 3u; mm[j] = 1u; continue; }   // (timing only, wrong frames: a prefetch without loads — what is the prefetch's latency worth?)
#endif
            hh[j] = col_h[c];
            cc[j] = (uint32_t)col_c[c];
            mm[j] = mask != nullptr ? (uint32_t)mask[c / (unsigned)p.N] : 1u;   // (wave-uniform branch; the division only with a mask)
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const bool valid = j <= touched && col + (unsigned)j <= last_col && mm[j] != 0u;
            const uint32_t pad = (uint32_t)column_padding(Hc, hh[j]);
            if (j <= touched && !valid) all_valid = false;
#ifdef RCW_DEV_SWITCHES
            if (p.fill_pairs == 2) { desc[lane * KS + j] = make_uint4(pad, (uint32_t)Hc - pad, 0x808080u + cc[j], valid ? 1u : 0u); continue; }
#endif
            desc[lane * KS + j] = make_uint4(pad, (uint32_t)Hc - pad, p.colour[cc[j] & 3], valid ? 1u : 0u);
        }
        const int state_l = (exists ? 1 : 0) | (all_valid ? 2 : 0);
        const int rem_l = (int)rem;
        col += dq; rem += dr;
        if (rem >= (unsigned)Hc) { rem -= (unsigned)Hc; col += 1; }
        __builtin_amdgcn_wave_barrier();                                     // (the lanes of a wavefront exchange through LDS: no reordering across)
#ifdef RCW_TRACE_WAVES
        {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && g < 1024 && grp < 18) { g_wave_trace[(g * 20 + grp) * 2] = t0; g_wave_trace[(g * 20 + grp) * 2 + 1] = t1; }
            grp += 1;
        }
#endif
        u32x4* dst = out4 + base * 64 + (unsigned long long)half * dstep;    // wave-uniform
        if (__ballot(state_l == 3) == ~0ull) {
            // every chunk of the group is whole and unmasked: no branch in the loop, the next chunk's pair(s) on their way
            int rel, r, rel_n, r_n;
            flat_locate(L, __builtin_amdgcn_readlane(rem_l, half), Hc, rel, r);
            uint4 d0 = desc[half * KS + rel], d1 = ALIGNED ? d0 : desc[half * KS + rel + 1];
            // (measured, µs per GiB: four-pixel groups inside one column 162 / 163 / 173 / 182 unrolled by 1 / 2 / 4 / 8 — unrolled, the
            // compiler bunches the stores of several chunks together, and the memory system takes evenly spaced stores best —;
            // groups that straddle columns, with their longer arithmetic, 182 / 177 / 171 / 171)
#pragma unroll (ALIGNED ? 1 : 4)
            for (int t = half; t < 64; t += STEP, dst += STEP * dstep) {
                // (the last trip fetches a 65th chunk's pair: lane 0's row again, and whatever lies behind in LDS; unused)
                flat_locate(L, __builtin_amdgcn_readlane(rem_l, t + STEP), Hc, rel_n, r_n);
                const uint4 n0 = desc[(t + STEP) * KS + rel_n], n1 = ALIGNED ? n0 : desc[(t + STEP) * KS + rel_n + 1];
                bool ok[4];
                const u32x4 v = flat_fill_pixels<ALIGNED>(r, Hc, d0, d1, ceil_c, floor_c, ok);
                __builtin_nontemporal_store(v, dst + lane);
                d0 = n0; d1 = n1; r = r_n;
            }
        } else {
            // The group holds a masked agent's border or the batch's end: its LEADING whole chunks (all of them up to the
            // batch's last chunk, in the last group of every wavefront) still take the branch-free loop — left to the general
            // loop alone, the last group cost a launch up to 8 µs (the wavefronts that end last: profiles/r03_fill_flat_wave_trace.txt)
            const unsigned long long whole = __ballot(state_l == 3);
            const int n_fast = (int)__builtin_ctzll(~whole);                 // (not all ones here)
            if (n_fast >= 4) {
                int rel, r, rel_n, r_n;
                flat_locate(L, __builtin_amdgcn_readlane(rem_l, half), Hc, rel, r);
                uint4 d0 = desc[half * KS + rel], d1 = ALIGNED ? d0 : desc[half * KS + rel + 1];
#pragma unroll 1
                for (int t = half; t < n_fast; t += STEP, dst += STEP * dstep) {
                    flat_locate(L, __builtin_amdgcn_readlane(rem_l, t + STEP), Hc, rel_n, r_n);
                    const uint4 n0 = desc[(t + STEP) * KS + rel_n], n1 = ALIGNED ? n0 : desc[(t + STEP) * KS + rel_n + 1];
                    bool ok[4];
                    const u32x4 v = flat_fill_pixels<ALIGNED>(r, Hc, d0, d1, ceil_c, floor_c, ok);
                    __builtin_nontemporal_store(v, dst + lane);
                    d0 = n0; d1 = n1; r = r_n;
                }
            }
            int t_first = n_fast >= 4 ? n_fast : 0;
            if (PAIR) { t_first += ((t_first ^ half) & 1); dst = out4 + base * 64 + (unsigned long long)t_first * dstep; }   // this wavefront's next chunk of the group
#pragma unroll 2
            for (int t = t_first; t < 64; t += STEP, dst += STEP * dstep) {
                const int s_state = __builtin_amdgcn_readlane(state_l, t);
                if (!(s_state & 1)) continue;                                // wave-uniform: past the end
                int rel, r;
                flat_locate(L, __builtin_amdgcn_readlane(rem_l, t), Hc, rel, r);
                const uint4 d0 = desc[t * KS + rel], d1 = ALIGNED ? d0 : desc[t * KS + rel + 1];
                bool ok[4];
                const u32x4 v = flat_fill_pixels<ALIGNED>(r, Hc, d0, d1, ceil_c, floor_c, ok);
                if (s_state & 2) {
                    __builtin_nontemporal_store(v, dst + lane);
                } else {                                                     // a masked agent's border / the batch's last chunk
                    const unsigned long long px0 = ((base + (unsigned long long)t * G) << 8) + 4u * (unsigned)lane;
                    uint32_t* const o = out + px0;
                    if (px0 + 0 < total_px && ok[0]) o[0] = v.x;
                    if (px0 + 1 < total_px && ok[1]) o[1] = v.y;
                    if (px0 + 2 < total_px && ok[2]) o[2] = v.z;
                    if (px0 + 3 < total_px && ok[3]) o[3] = v.w;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef RCW_TRACE_WAVES
    if (lane == 0 && g < 1024) {
        g_wave_trace[(g * 20 + 19) * 2] = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        g_wave_trace[(g * 20 + 18) * 2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
#endif
}

// Any other H_cam: one workgroup per agent.  The agent's N descriptors are turned
// into (padding, colour) pairs in LDS once; then the frame is streamed out, lanes along the flat pixel order (the
// image's contiguous axis), 16 bytes per lane when H_cam % 4 == 0 (four pixels never straddle a column then), the
// column of a store found with a Float32 reciprocal instead of an integer division.  A frame-per-workgroup stream
// reaches ≈ 70-75 % of the HBM peak on this chip (tools/fill_bench.hip, shape A) against 86 % for the moving window
// above — and against 14 % for the grid-stride kernel below, which did a 64-bit division per store and is kept
// only for frames too large for this one (N > 8192 columns or more than 2^25 pixels).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void rcw_fill_frame_kernel(const RcwDev p,
                                                                const int32_t* __restrict__ col_h,
                                                                const uint8_t* __restrict__ col_c,
                                                                uint32_t* __restrict__ out,
                                                                const uint8_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int a = blockIdx.x;
    if (mask != nullptr && mask[a] == 0) return;
    const int tid = threadIdx.x, N = p.N, Hc = p.Hc;
    int* const s_pad = reinterpret_cast<int*>(lds);         // [N] rows of ceiling (= rows of floor) of each column
    uint32_t* const s_col = lds + N;                        // [N] the column's colour
    for (int k = tid; k < N; k += kBlock) {
        s_pad[k] = column_padding(Hc, col_h[(size_t)a * N + k]);
        s_col[k] = p.colour[col_c[(size_t)a * N + k] & 3];
    }
    __syncthreads();
    uint32_t* const frame = out + (size_t)a * N * Hc;
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    if (VEC) {
        const int vpc = Hc >> 2, total = N * vpc;
        const float inv = 1.0f / (float)vpc;
        u32x4* const o4 = reinterpret_cast<u32x4*>(frame);
#pragma unroll 2
        for (int v = tid; v < total; v += kBlock) {
            const int col = fast_div(v, vpc, inv), r0 = (v - col * vpc) * 4;
            const int pad = s_pad[col];
            const uint32_t c = s_col[col];
            u32x4 px;
            px.x = pixel(r0 + 0, pad, Hc, c, ceil_c, floor_c);
            px.y = pixel(r0 + 1, pad, Hc, c, ceil_c, floor_c);
            px.z = pixel(r0 + 2, pad, Hc, c, ceil_c, floor_c);
            px.w = pixel(r0 + 3, pad, Hc, c, ceil_c, floor_c);
            o4[v] = px;
        }
    } else {
        const int total = N * Hc;
        const float inv = 1.0f / (float)Hc;
        for (int v = tid; v < total; v += kBlock) {
            const int col = fast_div(v, Hc, inv), r = v - col * Hc;
            frame[v] = pixel(r, s_pad[col], Hc, s_col[col], ceil_c, floor_c);
        }
    }
}

// The fallback for frames the kernel above cannot take: a grid-stride loop over the flat pixel array, every lane
// looks its own column up.  VEC: 16-byte stores (H_cam % 4 == 0).
template <bool VEC>
__global__ __launch_bounds__(kBlock) void rcw_fill_any_kernel(const RcwDev p,
                                                              const int32_t* __restrict__ col_h,
                                                              const uint8_t* __restrict__ col_c,
                                                              uint32_t* __restrict__ out, long long total_cols,
                                                              const uint8_t* __restrict__ mask)
{
    const long long per_col = VEC ? (p.Hc >> 2) : p.Hc;         // store units per column
    const long long total = total_cols * per_col;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += stride) {
        const long long c = idx / per_col;
        if (mask != nullptr && mask[c / p.N] == 0) continue;
        const int r0 = (int)(idx - c * per_col) * (VEC ? 4 : 1);
        const int pad = column_padding(p.Hc, col_h[c]);
        const uint32_t colour = p.colour[col_c[c] & 3];
        if (VEC) {
            u32x4 v;
            v.x = pixel(r0 + 0, pad, p.Hc, colour, p.ceiling_color, p.floor_color);
            v.y = pixel(r0 + 1, pad, p.Hc, colour, p.ceiling_color, p.floor_color);
            v.z = pixel(r0 + 2, pad, p.Hc, colour, p.ceiling_color, p.floor_color);
            v.w = pixel(r0 + 3, pad, p.Hc, colour, p.ceiling_color, p.floor_color);
            reinterpret_cast<u32x4*>(out)[idx] = v;
        } else {
            out[idx] = pixel(r0, pad, p.Hc, colour, p.ceiling_color, p.floor_color);
        }
    }
}

// ---- small kernels ---------------------------------------------------------------------------
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
    store_reward(p, a, false);
    p.done[a] = 0;
    p.status[a] = 0;
}

template <typename T>
__global__ void rcw_reset_kernel(const RcwDev p, const uint8_t* __restrict__ mask)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= p.B) return;
    if (mask != nullptr && mask[a] == 0) return;
    reset_agent<T>(p, a, p.tile_map + (size_t)a * p.nwords, nullptr);
}

// inject post-reset state: SR:118-132 with caller-chosen draws
template <typename T>
__global__ void rcw_set_state_kernel(const RcwDev p, const int2* __restrict__ goal,
                                     const typename Real<T>::vec2* __restrict__ pos, const int32_t* __restrict__ dir,
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
    Real<T>::pos(p)[a] = pos[a];                  // SR:126
    p.dir[a] = dir[a];                            // SR:129
    store_reward(p, a, false);                    // SR:131
    p.done[a] = 0;                                // SR:132
}

// cast_rays!(world) SR:195-231 with the ray buffers materialised (rcw_rays)
template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_rays_kernel(const RcwDev p, int first, RcwRayOut out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int local = blockIdx.x;
    const int a = first + local;
    const int tid = threadIdx.x;
    uint8_t* tb = reinterpret_cast<uint8_t*>(lds);
    stage_tile_bytes(tb, p.tile_map + (size_t)a * p.nwords, p.H * p.W, tid, kBlock);
    __syncthreads();
    const typename Real<T>::vec2 pos = Real<T>::pos(p)[a];
    const int d = p.dir[a];
    const T* tab = Real<T>::ray_table(p) + (size_t)d * RCW_TABLE_ROWS * p.N;
    T* out_dist = static_cast<T*>(out.dist);
    T* out_dirs = static_cast<T*>(out.dirs);
    for (int i = tid; i < p.N; i += kBlock) {
        const T dx = tab[i], dy = tab[p.N + i];
        const RayHit<T> r = cast_ray<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, pos.x, pos.y, dx, dy, tab[2 * p.N + i],
                                                          tab[3 * p.N + i]);
        const int hit_j = r.t / p.H, hit_i = r.t - hit_j * p.H;           // 0-based stop tile
        const size_t q = (size_t)local * p.N + i;
        if (out.stop_ij) { out.stop_ij[2 * q] = r.oob ? 1 : hit_i + 1; out.stop_ij[2 * q + 1] = r.oob ? 1 : hit_j + 1; }
        if (out.hit_dim) out.hit_dim[q] = r.oob ? 0 : r.dim;
        if (out_dist) out_dist[q] = r.oob ? (T)0 : r.dist;
        if (out_dirs) { out_dirs[2 * q] = dx; out_dirs[2 * q + 1] = dy; }
    }
}

// ---- update_top_view!(env)  SR:446-483 (+ draw_tile_map! SR:342-372) ----------------------------
// The reference's debug view: tile squares with a grid, one line per ray, the player circle.  Not
// an observation and off by default (cfg.render_top_view).  The line and circle rasterisers are
// SimpleDraw 0.3's (un-vendored): Bresenham and the midpoint circle are ASSUMED — parity unpinned.
//
// Two kernels.  rcw_top_view_kernel (below) is the one that runs whenever the image's bit planes fit
// in LDS: it writes every pixel exactly once.  rcw_top_view_inplace_kernel (this one) is the fallback
// for larger images: three phases separated by barriers, later phases overwriting pixels of
// earlier ones in HBM exactly as the reference does.
__device__ __forceinline__ void put_pixel(uint32_t* img, int Ht, int Wt, int i, int j, uint32_t c)
{
    if (i >= 1 && i <= Ht && j >= 1 && j <= Wt) img[(size_t)(i - 1) + (size_t)Ht * (j - 1)] = c;
}
template <typename T>
__device__ __forceinline__ int wu_to_pu(T x, int pu) { return (int)rfloor(x * (T)pu) + 1; }   // UT:6

__device__ __forceinline__ uint32_t top_view_tile_pixel(const uint8_t* tb, int H, int pu, int ip0, int jp0)
{
    const int i = ip0 / pu, j = jp0 / pu;                   // 0-based tile
    const int ri = ip0 - i * pu, rj = jp0 - j * pu;
    if (ri == 0 || ri == pu - 1 || rj == 0 || rj == pu - 1) return 0x00ccccccu;   // SR:364-367
    const uint32_t bits = tb[i + H * j];
    return (bits & 1u) ? 0x00FFFFFFu : ((bits & 2u) ? 0x00FF0000u : 0x00000000u); // findfirst SR:355-360, colours SR:288
}

template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_top_view_inplace_kernel(const RcwDev p, const uint8_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int a = blockIdx.x;
    const int tid = threadIdx.x;
    if (mask != nullptr && mask[a] == 0) return;
    uint8_t* tb = reinterpret_cast<uint8_t*>(lds);
    stage_tile_bytes(tb, p.tile_map + (size_t)a * p.nwords, p.H * p.W, tid, kBlock);
    __syncthreads();
    const int pu = p.pu, Ht = p.H * pu, Wt = p.W * pu;
    uint32_t* img = p.top_view + (size_t)a * Ht * Wt;

    // ---- draw_tile_map!: each tile's square and its one-pixel frame.  Tiles do not overlap, so the
    // reference's tile-by-tile order does not matter; rows of a column are contiguous (column-major).
    if ((pu & 3) == 0) {
        // One wavefront per image column (its tile column j and frame flag are wave-uniform), lanes
        // along the contiguous rows, four pixels per lane — they never straddle a tile when pu % 4 == 0.
        const int wave = tid >> 6, lane = tid & 63;
        u32x4* out = reinterpret_cast<u32x4*>(img);
        for (int jp0 = wave; jp0 < Wt; jp0 += kBlock / 64) {
            const int j = jp0 / pu;
            const int rj = jp0 - j * pu;
            const bool frame_col = rj == 0 || rj == pu - 1;                  // SR:366-367
            for (int ip0 = lane * 4; ip0 < Ht; ip0 += 256) {
                const int i = ip0 / pu;
                const int ri = ip0 - i * pu;
                const uint32_t bits = tb[i + p.H * j];
                const uint32_t fill = (bits & 1u) ? 0x00FFFFFFu : ((bits & 2u) ? 0x00FF0000u : 0x00000000u);   // SR:355-360
                const uint32_t inner = frame_col ? 0x00ccccccu : fill;
                u32x4 v;
                v.x = ri == 0 ? 0x00ccccccu : inner;                         // SR:364: first row of the tile
                v.y = inner;
                v.z = inner;
                v.w = ri + 3 == pu - 1 ? 0x00ccccccu : inner;                // SR:365: last row of the tile
                out[(size_t)jp0 * (Ht >> 2) + (ip0 >> 2)] = v;
            }
        }
    } else if ((Ht & 3) == 0) {
        const int vpc = Ht >> 2;
        u32x4* out = reinterpret_cast<u32x4*>(img);
        for (int idx = tid; idx < vpc * Wt; idx += kBlock) {
            const int jp0 = idx / vpc, ip0 = (idx - jp0 * vpc) * 4;
            u32x4 v;
            v.x = top_view_tile_pixel(tb, p.H, pu, ip0 + 0, jp0);
            v.y = top_view_tile_pixel(tb, p.H, pu, ip0 + 1, jp0);
            v.z = top_view_tile_pixel(tb, p.H, pu, ip0 + 2, jp0);
            v.w = top_view_tile_pixel(tb, p.H, pu, ip0 + 3, jp0);
            out[idx] = v;
        }
    } else {
        for (int idx = tid; idx < Ht * Wt; idx += kBlock) {
            const int jp0 = idx / Ht;
            img[idx] = top_view_tile_pixel(tb, p.H, pu, idx - jp0 * Ht, jp0);
        }
    }
    __syncthreads();   // (waits for the stores above: the lines below overwrite some of those pixels)

    // ---- one line per ray from the player to the ray's stop point  SR:473-477 ----
    const typename Real<T>::vec2 pos = Real<T>::pos(p)[a];
    const int d = p.dir[a];
    const int ip = wu_to_pu<T>(pos.x, pu), jp = wu_to_pu<T>(pos.y, pu);      // SR:468
    const T* tab = Real<T>::ray_table(p) + (size_t)d * RCW_TABLE_ROWS * p.N;
    for (int i = tid; i < p.N; i += kBlock) {
        const T dx = tab[i], dy = tab[p.N + i];
        const RayHit<T> r = cast_ray<T, TIE_LE, DIST_PRE>(tb, p.H, p.W, pos.x, pos.y, dx, dy, tab[2 * p.N + i],
                                                          tab[3 * p.N + i]);
        const T dist = r.oob ? (T)0 : r.dist;
        const T ox = dist * dx, oy = dist * dy;                              // ray_distance_wu * ray_direction_wu
        const T ex = pos.x + ox, ey = pos.y + oy;
        int i1 = ip, j1 = jp;
        const int i2 = wu_to_pu<T>(ex, pu), j2 = wu_to_pu<T>(ey, pu);
        // SD.Line: Bresenham, all octants, both end points (assumed)
        const int di = abs(i2 - i1), dj = -abs(j2 - j1);
        const int si = i1 < i2 ? 1 : -1, sj = j1 < j2 ? 1 : -1;
        int err = di + dj;
        for (int guard = 0; guard <= Ht + Wt + 4 * pu; ++guard) {           // a line has at most di - dj + 1 pixels
            put_pixel(img, Ht, Wt, i1, j1, 0x00808080u);                     // ray_color SR:289
            if (i1 == i2 && j1 == j2) break;
            const int e2 = 2 * err;
            if (e2 >= dj) { err += dj; i1 += si; }
            if (e2 <= di) { err += di; j1 += sj; }
        }
    }
    __syncthreads();

    // ---- the player: SD.Circle(Point(ip - rp, jp - rp), 2 rp + 1)  SR:480 (midpoint circle, assumed) ----
    if (tid == 0) {
        const int rp = p.top_rp;                                             // SR:469
        int x = 0, y = rp, dd = 1 - rp;
        while (x <= y) {
            put_pixel(img, Ht, Wt, ip + x, jp + y, 0x00c0c0c0u); put_pixel(img, Ht, Wt, ip - x, jp + y, 0x00c0c0c0u);
            put_pixel(img, Ht, Wt, ip + x, jp - y, 0x00c0c0c0u); put_pixel(img, Ht, Wt, ip - x, jp - y, 0x00c0c0c0u);
            put_pixel(img, Ht, Wt, ip + y, jp + x, 0x00c0c0c0u); put_pixel(img, Ht, Wt, ip - y, jp + x, 0x00c0c0c0u);
            put_pixel(img, Ht, Wt, ip + y, jp - x, 0x00c0c0c0u); put_pixel(img, Ht, Wt, ip - y, jp - x, 0x00c0c0c0u);
            x += 1;
            if (dd < 0) dd += 2 * x + 1;
            else { y -= 1; dd += 2 * (x - y) + 1; }
        }
    }
}


// ---- the write-once top view -----------------------------------------------------------------------
// Every pixel of the (H·pu, W·pu) image is stored exactly once, and drawing overlaps with storing:
// a workgroup has 8 wavefronts in two groups of four and walks through its agents in slots.  In slot s
//   * the DRAW group rasterises agent s into one of two LDS buffers: the agent's tile map (a byte per
//     tile), a `line` bit plane (one bit per pixel, bit index (j-1)·top_col_bits + (i-1))
//     and a `circ` plane for the 2·rp+1 image columns around the player.  One lane per ray: cast (the same
//     DDA as the camera path), end point SR:476, then the line's pixels are OR-ed into `line` with LDS
//     atomics.  All lines start at the player's pixel and neighbouring rays share most of their first
//     pixels, so a lane whose left neighbour is on the same pixel at the same step leaves the bit to it;
//   * the STORE group streams agent s-1 out of the other buffer: lanes along the contiguous axis (rows of a
//     column), 16 bytes per lane, colour = circle > ray line > tile frame > tile fill — the reference's
//     overwrite order (SR:362-367 fill then frame per tile, SR:473-477 lines, SR:480 circle) per pixel.
// Two workgroup barriers per slot (LDS only: stores stay in flight across them).  The drawing is VALU/LDS
// work, the storing is HBM work; run one after the other they add up (measured: 239 µs at 4096 x 256² px),
// overlapped the kernel approaches the store time.  The grid is persistent (4 workgroups per CU).
// Algorithmic bytes: 4·(H·pu)·(W·pu) per agent, the HBM write roofline bounds it like the camera fill.
//
// SD.Line (ASSUMED Bresenham, all octants, both end points, as in the in-place kernel above): with a = the
// longer and b = the shorter extent, pixel k = 0..a of the line sits k steps along the major axis and
// floor((2·b·k + a) / (2·a)) steps along the minor axis — the closed form of the error recurrence
// `e2 = 2 err; if e2 >= dj ...; if e2 <= di ...` (checked exhaustively against it on the CPU,
// tests/test_host_logic.py).  The loop carries the remainder of that division instead of the error term.
constexpr int kTopBlock = 512;          // 4 draw + 4 store wavefronts
constexpr int kTopGroup = 256;
constexpr int kTopDummyWords = 64;      // where lanes with nothing to draw aim their (harmless) atomic

// A plane stores image column j (Ht pixels, contiguous in the image) at bit offset j * top_col_bits: the column
// stride is padded to an ODD number of words, so that the pixels of one wavefront step — which lie on an arc across
// neighbouring columns when the agent looks along the rows — fall into different LDS banks (an unpadded 256-pixel
// column is 8 words: neighbouring columns would share only 4 banks).
__host__ __device__ __forceinline__ int top_col_bits(const RcwDev& p)
{
    const int words = (p.H * p.pu + 31) / 32;
    return 32 * (words | 1);
}
__host__ __device__ __forceinline__ size_t top_line_words(const RcwDev& p)
{
    return ((size_t)p.W * p.pu * (top_col_bits(p) / 32) + 3) & ~(size_t)3;     // multiple of 4 words
}
__host__ __device__ __forceinline__ size_t top_circ_words(const RcwDev& p)
{
    return ((size_t)(2 * p.top_rp + 1) * (top_col_bits(p) / 32) + 3) & ~(size_t)3;
}
__host__ __device__ __forceinline__ size_t top_tile_words(const RcwDev& p) { return (((size_t)p.H * p.W + 15) & ~(size_t)15) / 4; }
// one buffer: [header 4 words | tile bytes | line | circ | dummy]
__host__ __device__ __forceinline__ size_t top_buf_words(const RcwDev& p)
{
    return 4 + top_tile_words(p) + top_line_words(p) + top_circ_words(p) + kTopDummyWords;
}

struct TopBuf {
    int* hdr;           // [0] ip, [1] jp: the player's pixel (1-based)  SR:468
    uint8_t* tb;        // [H*W] tile bytes
    uint32_t* line;
    uint32_t* circ;
    uint32_t* dummy;
};
__device__ __forceinline__ TopBuf top_buf(const RcwDev& p, uint32_t* base)
{
    TopBuf b;
    b.hdr = reinterpret_cast<int*>(base);
    b.tb = reinterpret_cast<uint8_t*>(base + 4);
    b.line = base + 4 + top_tile_words(p);
    b.circ = b.line + top_line_words(p);
    b.dummy = b.circ + top_circ_words(p);
    return b;
}

// LDS-only workgroup barrier: waits for this wavefront's LDS operations, not for its global stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t tile_fill_colour(uint32_t bits)
{
    return (bits & 1u) ? 0x00FFFFFFu : ((bits & 2u) ? 0x00FF0000u : 0x00000000u);       // findfirst SR:355-360, colours SR:288
}

// draw group, first half of a slot: stage the agent's tile map, clear the planes
__device__ __forceinline__ void top_prepare(const RcwDev& p, int a, const TopBuf& b, int tid, int group = kTopGroup)
{
    stage_tile_bytes(b.tb, p.tile_map + (size_t)a * p.nwords, p.H * p.W, tid, group);
    u32x4* z = reinterpret_cast<u32x4*>(b.line);
    const int nz = (int)((top_line_words(p) + top_circ_words(p)) >> 2);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int k = tid; k < nz; k += group) z[k] = zero;
}

#ifdef RCW_TRACE_WAVES
// Measurement build only (tools/draw_trace.py): the first wavefront of the draw workgroups of agents 0..2047 leaves s_memrealtime at
// entry | planes cleared, barrier | rays cast, lines set up | lines walked | barrier | planes copied out and acknowledged, and where it ran
__device__ unsigned long long g_draw_trace[2048 * 20];
}  // namespace
extern "C" __attribute__((visibility("default"))) int rcw_draw_trace_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_draw_trace), sizeof(unsigned long long) * 2048 * 20);
}
namespace {
#define RCW_DRAW_STAMP(k) do { if (tid == 0 && a < 2048) g_draw_trace[a * 20 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RCW_DRAW_STAMP(k) do { } while (0)
#endif

// draw group, second half: one line per ray from the player to the ray's stop point (SR:473-477) and the player
template <typename T, bool TIE_LE, bool DIST_PRE>
__device__ __forceinline__ void top_draw(const RcwDev& p, int a, const TopBuf& b, int tid, bool with_circle = true, int group = kTopGroup)
{
    const int pu = p.pu, Ht = p.H * pu, Wt = p.W * pu, rp = p.top_rp;
    const typename Real<T>::vec2 pos = Real<T>::pos(p)[a];
    const int d = p.dir[a];
    const int ip = wu_to_pu<T>(pos.x, pu), jp = wu_to_pu<T>(pos.y, pu);      // SR:468 (1-based)
    if (tid == 0) { b.hdr[0] = ip; b.hdr[1] = jp; }
    const T* tab = Real<T>::ray_table(p) + (size_t)d * RCW_TABLE_ROWS * p.N;
    const bool start_inside = ip >= 1 && ip <= Ht && jp >= 1 && jp <= Wt;
    const int cb_ = top_col_bits(p);
    uint32_t* const dummy = b.dummy + (tid & 63);
    // Lanes per ray: with fewer rays than lanes (N <= group / 2) a line is cut into `parts` segments, one lane each
    // (lanes of one ray are group / parts apart, so a wavefront holds neighbouring rays' same segment).
    const int npad = (p.N + 63) & ~63;
    int parts = 1;
    while (parts * 2 * npad <= group) parts *= 2;
    const int rays_per_pass = group / parts;                                 // a multiple of 64
    const int part = tid / rays_per_pass, ray_in_pass = tid - part * rays_per_pass;
    for (int i0 = 0; i0 < p.N; i0 += rays_per_pass) {
        const int i = i0 + ray_in_pass;
        // (bit addresses below are relative to the start of the `line` plane; the dummy words lie behind it and `circ`)
        const int dummy_bit = (int)((dummy - b.line) * 32);
        int n = 0, addr0 = dummy_bit, step_maj = 0, step_both = 0;
        uint32_t slope = 0u;
        bool checked = false;
        int i2 = ip, j2 = jp;
        if (i < p.N) {
            const T dx = tab[i], dy = tab[p.N + i];
            const RayHit<T> r = cast_ray<T, TIE_LE, DIST_PRE>(b.tb, p.H, p.W, pos.x, pos.y, dx, dy, tab[2 * p.N + i],
                                                              tab[3 * p.N + i]);
            const T dist = r.oob ? (T)0 : r.dist;
            const T ox = dist * dx, oy = dist * dy;                          // ray_distance_wu * ray_direction_wu
            const T ex = pos.x + ox, ey = pos.y + oy;
            i2 = wu_to_pu<T>(ex, pu); j2 = wu_to_pu<T>(ey, pu);             // SR:476
            // a line whose end points are both on the image stays on it; anything else takes the clipped walk
            checked = !(start_inside && i2 >= 1 && i2 <= Ht && j2 >= 1 && j2 <= Wt);
            if (checked) {
                checked = part == 0;                                         // (one lane of the ray takes the clipped walk)
            } else {
                const int di = abs(i2 - ip), dj = abs(j2 - jp);
                const int si = ip < i2 ? 1 : -1, sj = jp < j2 ? cb_ : -cb_;    // steps of the plane's bit index
                const bool imaj = di >= dj;
                const int la = imaj ? di : dj, lb = imaj ? dj : di;
                n = la + 1;
                step_maj = imaj ? si : sj;
                step_both = si + sj;
                addr0 = (jp - 1) * cb_ + (ip - 1);
                // floor(2^32 · b / a), exact in Float64 (b · 2^32 is exact, the quotient's fraction is a multiple of 1/a)
                slope = lb >= la ? 0xFFFFFFFFu : (uint32_t)((double)lb * 4294967296.0 / (double)la);
            }
        }
        // Pixel k of the line sits k steps along the major axis and floor(k·b/a + 1/2) along the minor one (see above).
        // The loop carries the FRACTION of k·slope/2^32 + 1/2 + 2^-18 in 32 bits and steps the minor axis on its
        // carry: with slope/2^32 in (b/a - 2^-32, b/a] the carried value exceeds the true one by less than 2^-18 and
        // by more than 0 for k < 2^14, and the true value's fraction is a multiple of 1/(2a) > 2^-18 — so no integer
        // lies between them and the floors agree (lines on an image whose bit plane fits in LDS have a < 2^12; the
        // closed form against the error-term walk and the carry against the closed form: tests/test_host_logic.py).
        // Every lane walks its whole line but starts somewhere along it and wraps round: walked in step from the
        // player, the 64 neighbouring rays of a wavefront sit on one small arc at every step — the same plane word or
        // two for the first dozens of steps, and same-word LDS atomics serialise (a third of the draw kernel's
        // wave-cycles waited for the LDS queue, SQ_WAIT_INST_LDS).  Neighbouring lanes start 37/64 of a line apart.
        // A lane that is through before the wavefront's longest line simply goes round again (OR is idempotent); a
        // lane without a line ORs into a private dummy word.  No divergent branch in the loop.
        const uint32_t frac0 = 0x80000000u + (1u << 14);
        const int ks = (int)(((long long)part * n) / parts), ke = (int)(((long long)(part + 1) * n) / parts);   // this lane's pixels of the line
        const int len = ke - ks;
        if (len == 0) { addr0 = dummy_bit; step_maj = step_both = 0; slope = 0u; }
        const int k0 = len > 1 ? ks + (int)((((unsigned)(tid * 37) & 63u) * (unsigned)len) >> 6) : ks;
        const unsigned long long at_ks = (unsigned long long)(unsigned)ks * slope + frac0;     // v_mad_u64_u32
        const unsigned long long at_k0 = (unsigned long long)(unsigned)k0 * slope + frac0;
        const uint32_t frac_s = (uint32_t)at_ks;
        const int addr_s = addr0 + ks * step_maj + (int)(at_ks >> 32) * (step_both - step_maj);
        uint32_t frac = (uint32_t)at_k0;
        int addr = addr0 + k0 * step_maj + (int)(at_k0 >> 32) * (step_both - step_maj);
        int rem = len > 0 ? ke - k0 : 0x7fffffff;                            // steps until the wrap
        int nmax = len;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        nmax = __builtin_amdgcn_readfirstlane(nmax);                         // (the trip count is the wavefront's longest segment: a scalar loop)
        char* const plane = reinterpret_cast<char*>(b.line);
        if (i0 == 0) { RCW_DRAW_STAMP(2); }
#ifdef RCW_TRACE_WAVES
        if (i0 == 0 && tid == 0 && a < 2048) g_draw_trace[a * 20 + 8] = (unsigned long long)nmax;
#endif
        for (int k = 0; k < nmax; ++k) {
            uint32_t* const w = reinterpret_cast<uint32_t*>(plane + (((unsigned)addr >> 3) & ~3u));
            __hip_atomic_fetch_or(w, 1u << (addr & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t next = frac + slope;
            addr += next < frac ? step_both : step_maj;                      // carry: a step along the minor axis too
            frac = next;
            rem -= 1;
            if (__ballot(rem == 0) != 0ull) {                                // some lane is back at the start of its segment
                asm volatile("" ::: "memory");                               // (keeps this a branch: if-converted, its selects run every step)
                const bool wrap = rem == 0;
                rem = wrap ? len : rem; frac = wrap ? frac_s : frac; addr = wrap ? addr_s : addr;
            }
        }
        if (__ballot(checked)) {
            // clipped walk (SimpleDraw skips pixels off the image): the error-term loop as written
            if (checked) {
                int i1 = ip, j1 = jp;
                const int di = abs(i2 - i1), dj = -abs(j2 - j1);
                const int si = i1 < i2 ? 1 : -1, sj = j1 < j2 ? 1 : -1;
                int err = di + dj;
                for (long long guard = 0; guard <= (long long)di - dj; ++guard) {
                    if (i1 >= 1 && i1 <= Ht && j1 >= 1 && j1 <= Wt) {
                        const int q = (j1 - 1) * cb_ + (i1 - 1);
                        __hip_atomic_fetch_or(b.line + (q >> 5), 1u << (q & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (i1 == i2 && j1 == j2) break;
                    const int e2 = 2 * err;
                    if (e2 >= dj) { err += dj; i1 += si; }
                    if (e2 <= di) { err += di; j1 += sj; }
                }
            }
        }
    }
    // the player: SD.Circle(Point(ip - rp, jp - rp), 2 rp + 1)  SR:480 (midpoint circle, assumed).  Its plane is
    // separate from the lines', so one lane of the LAST wavefront draws it while the others finish their lines.
    if (with_circle && tid == group - 1) {
        const int jc0 = jp - rp;                                             // first image column of the circle plane (1-based)
        int x = 0, y = rp, dd = 1 - rp;
        auto put = [&](int i, int j) {
            if (i >= 1 && i <= Ht && j >= 1 && j <= Wt) {
                const int q = (j - jc0) * cb_ + (i - 1);
                b.circ[q >> 5] |= 1u << (q & 31);
            }
        };
        while (x <= y) {
            put(ip + x, jp + y); put(ip - x, jp + y); put(ip + x, jp - y); put(ip - x, jp - y);
            put(ip + y, jp + x); put(ip - y, jp + x); put(ip + y, jp - x); put(ip - y, jp - x);
            x += 1;
            if (dd < 0) dd += 2 * x + 1;
            else { y -= 1; dd += 2 * (x - y) + 1; }
        }
    }
}

// store group: agent a's image, every pixel once
__device__ __forceinline__ void top_store(const RcwDev& p, int a, const TopBuf& b, int tid)
{
    const int pu = p.pu, Ht = p.H * pu, Wt = p.W * pu, rp = p.top_rp;
    uint32_t* img = p.top_view + (size_t)a * Ht * Wt;
    const uint32_t ray_c = 0x00808080u, player_c = 0x00c0c0c0u, grid_c = 0x00ccccccu;   // SR:289-290, SR:364-367
    const float inv_pu = 1.0f / (float)pu;
    const int jc0 = b.hdr[1] - rp;
    const int box = 2 * rp;
    const int cbits = top_col_bits(p);
    const int c_lo = 0, c_hi = Wt;
    if ((pu & 3) == 0) {
        // One wavefront per image column (256 rows per pass; the lanes past the end of a shorter last pass idle), lanes along the contiguous rows, four pixels a lane:
        // they never straddle a tile.  What depends on the rows only (tile row, frame rows) is computed once per
        // row block, what depends on the column's tile once per tile, the column itself is wave-uniform (scalar
        // unit), and the overlay is skipped for a column none of whose 256 pixels carries a line or circle bit:
        // per column and lane that leaves one LDS read, a bit-field extract, four selects and the store.
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        u32x4* out = reinterpret_cast<u32x4*>(img);
        const int vpc = Ht >> 2;
        const int wpc = top_col_bits(p) >> 5;                               // plane words per (padded) column
        const int step = kTopGroup / 64;                                    // columns between two of this wavefront's
        [[maybe_unused]] const int ncols = (c_hi - c_lo - wave + step - 1) / step;           // wave-uniform trip count (development path below)
        auto overlay = [](uint32_t bits, int e, uint32_t colour, uint32_t under) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)bits, e, 1);   // 0 or ~0
            return (m & colour) | (~m & under);                             // v_bfi_b32
        };
        const int cpt = pu >> 2;                                            // this wavefront's columns per tile column
        const int lstep = step * wpc;
        const size_t dstep = (size_t)step * vpc;
        for (int r0 = 0; r0 < Ht; r0 += 256) {
            const bool active = r0 + lane * 4 < Ht;                          // (Ht % 256 != 0: a shorter last pass)
            const int ip0 = active ? r0 + lane * 4 : 0;
            const int ti = fast_div(ip0, pu, inv_pu), ri = ip0 - ti * pu;
            const bool first_row = ri == 0, last_row = ri + 3 == pu - 1;    // SR:364-365: the tile's frame rows
            const uint8_t* const tile_row = b.tb + ti;
            const int sh = ip0 & 31;
            int jp0 = wave;                                                  // this wavefront's columns: wave, wave + 4, ...
            const uint32_t* lp = b.line + jp0 * wpc + (ip0 >> 5);
            u32x4* dst = out + (size_t)jp0 * vpc + (ip0 >> 2);
#ifdef RCW_DEV_SWITCHES
            if (p.top_debug & 8) {      // development: the bare store stream of this path (no pixel logic, no LDS reads)
                for (int c = 0; c < ncols; ++c) { const u32x4 o = {grid_c, grid_c, grid_c, grid_c}; if (active) *dst = o; dst += dstep; }
                continue;
            }
#endif
            // Tile columns outermost: step = 4 divides pu, so every tile column holds cpt = pu / 4 of this wavefront's
            // columns, the tile's colour is read once (the next tile's byte is already on its way), and a frame
            // column (SR:366-367) can only be the first one (wavefront 0) or the last one (wavefront 3) of a tile.
            uint32_t tile_next = tile_row[0];
            for (int tj = 0; tj < p.W; ++tj) {
                const uint32_t fill = tile_fill_colour(tile_next);
                if (tj + 1 < p.W) tile_next = tile_row[p.H * (tj + 1)];
                const uint32_t fx = first_row ? grid_c : fill, fw = last_row ? grid_c : fill;
                constexpr int U = 4;
                for (int cc = 0; cc < cpt; cc += U) {
                    // U columns per trip: their plane words are read from LDS first, so that the stores that follow
                    // do not each wait for an LDS round trip
                    uint32_t words[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) words[u] = cc + u < cpt ? lp[u * lstep] : 0u;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (cc + u >= cpt) break;                            // wave-uniform
                        const bool frame_col = (wave == 0 && cc + u == 0) || (wave == step - 1 && cc + u == cpt - 1);
                        u32x4 o;
                        o.x = frame_col ? grid_c : fx;
                        o.y = frame_col ? grid_c : fill;
                        o.z = o.y;
                        o.w = frame_col ? grid_c : fw;
                        const uint32_t lb = __builtin_amdgcn_ubfe(words[u], sh, 4);
                        if (__ballot(lb != 0u) != 0ull) {                    // some pixel of this column is on a ray line
                            o.x = overlay(lb, 0, ray_c, o.x); o.y = overlay(lb, 1, ray_c, o.y);
                            o.z = overlay(lb, 2, ray_c, o.z); o.w = overlay(lb, 3, ray_c, o.w);
                        }
                        if ((unsigned)(jp0 + 1 - jc0) <= (unsigned)box) {    // wave-uniform: a column of the circle's box
                            const uint32_t cb = __builtin_amdgcn_ubfe(b.circ[(jp0 + 1 - jc0) * wpc + (ip0 >> 5)], sh, 4);
                            o.x = overlay(cb, 0, player_c, o.x); o.y = overlay(cb, 1, player_c, o.y);
                            o.z = overlay(cb, 2, player_c, o.z); o.w = overlay(cb, 3, player_c, o.w);
                        }
                        if (active) dst[(size_t)u * dstep] = o;   // plain, not non-temporal: 224 vs 237 us for the kernel (the opposite of the camera fill)
                        jp0 += step;
                    }
                    lp += U * lstep; dst += U * dstep;
                }
                // (a tile with cpt not a multiple of U advanced the pointers past its end)
                if (cpt % U) { lp -= (U - cpt % U) * lstep; dst -= (U - cpt % U) * dstep; }
            }
        }
    } else if ((Ht & 3) == 0) {
        // any pu, Ht % 4 == 0: four pixels per lane (they never straddle a column), tiles looked up per pixel
        const int vpc = Ht >> 2, v_lo = c_lo * vpc, v_hi = c_hi * vpc;
        const int qstep = kTopGroup / vpc, rstep = kTopGroup - qstep * vpc;
        int jp0 = (v_lo + tid) / vpc, rem = (v_lo + tid) - jp0 * vpc;        // column, vector within the column
        u32x4* out = reinterpret_cast<u32x4*>(img);
        for (int v = v_lo + tid; v < v_hi; v += kTopGroup) {
            const int ip0 = rem * 4;
            const int tj = fast_div(jp0, pu, inv_pu), rj = jp0 - tj * pu;
            const bool frame_col = rj == 0 || rj == pu - 1;
            uint32_t px[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ti = fast_div(ip0 + e, pu, inv_pu), ri = ip0 + e - ti * pu;
                px[e] = (frame_col || ri == 0 || ri == pu - 1) ? grid_c : tile_fill_colour(b.tb[ti + p.H * tj]);
            }
            const int lin = jp0 * cbits + ip0;
            const uint32_t lb = b.line[lin >> 5] >> (lin & 31);
            uint32_t cb = 0u;
            const int cj = jp0 + 1 - jc0;
            if ((unsigned)cj <= (unsigned)box) { const int q = cj * cbits + ip0; cb = b.circ[q >> 5] >> (q & 31); }
            u32x4 o;
            o.x = (cb & 1u) ? player_c : ((lb & 1u) ? ray_c : px[0]);
            o.y = (cb & 2u) ? player_c : ((lb & 2u) ? ray_c : px[1]);
            o.z = (cb & 4u) ? player_c : ((lb & 4u) ? ray_c : px[2]);
            o.w = (cb & 8u) ? player_c : ((lb & 8u) ? ray_c : px[3]);
            __builtin_nontemporal_store(o, out + v);
            jp0 += qstep; rem += rstep;
            if (rem >= vpc) { rem -= vpc; jp0 += 1; }
        }
    } else {
        const int v_lo = c_lo * Ht, v_hi = c_hi * Ht;
        const int qstep = kTopGroup / Ht, rstep = kTopGroup - qstep * Ht;
        int jp0 = (v_lo + tid) / Ht, ip0 = (v_lo + tid) - jp0 * Ht;
        for (int v = v_lo + tid; v < v_hi; v += kTopGroup) {
            const int tj = fast_div(jp0, pu, inv_pu), rj = jp0 - tj * pu;
            const int ti = fast_div(ip0, pu, inv_pu), ri = ip0 - ti * pu;
            uint32_t c = (rj == 0 || rj == pu - 1 || ri == 0 || ri == pu - 1) ? grid_c : tile_fill_colour(b.tb[ti + p.H * tj]);
            const int lin = jp0 * cbits + ip0;
            if ((b.line[lin >> 5] >> (lin & 31)) & 1u) c = ray_c;
            const int cj = jp0 + 1 - jc0;
            if ((unsigned)cj <= (unsigned)box) { const int q = cj * cbits + ip0; if ((b.circ[q >> 5] >> (q & 31)) & 1u) c = player_c; }
            img[v] = c;
            jp0 += qstep; ip0 += rstep;
            if (ip0 >= Ht) { ip0 -= Ht; jp0 += 1; }
        }
    }
}

// Hand-offs between wavefronts of one workgroup through counters in LDS that only ever grow: a wavefront adds one
// when its own LDS operations are done (`signal`), a waiter spins (with s_sleep) until the count it needs is there.
// No s_barrier in the steady state: a barrier per agent would make every agent cost max(draw, store), and the
// drawing time varies with the agent's ray lengths — with a ring of buffers the draw group runs ahead and only the
// averages have to balance.
__device__ __forceinline__ void lds_signal(int* counter)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait(int* counter, int target)
{
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}

template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kTopBlock, 6) void rcw_top_view_kernel(const RcwDev p, const uint8_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int role = threadIdx.x >> 8;        // 0: draw group, 1: store group (wave-uniform)
    const int tid = threadIdx.x & (kTopGroup - 1);
    const int G = gridDim.x;
    const int n = (p.B - (int)blockIdx.x + G - 1) / G;                      // agents of this workgroup: blockIdx.x + q*G
    const size_t bw = top_buf_words(p);
    // lds[0]: draw-group wavefronts that have prepared their current agent; lds[1]: ... that have finished drawing
    // (summed over agents); lds[2]: store-group wavefronts that have finished storing (summed over agents)
    int* const c_prepared = reinterpret_cast<int*>(lds);
    int* const c_drawn = c_prepared + 1;
    int* const c_stored = c_prepared + 2;
    uint32_t* const bufs = lds + 4;
    if (threadIdx.x < 4) c_prepared[threadIdx.x] = 0;
    lds_barrier();
    // A ring of K = p.top_lds buffers (1..3, as many as fit in LDS).  The draw group draws agent q into buffer
    // q mod K as soon as the store group has finished agent q - K; the store group stores agent q as soon as all four
    // draw wavefronts have finished it.  With K = 1 the two simply alternate.  The two groups run separate loops
    // (separate code paths): in a shared loop the compiler's wait-count bookkeeping for the draw group's loads also
    // drained the store group's stores once per agent.
    const int K = p.top_lds;
    if (role == 0) {
        int prepared = 0;
        for (int q = 0; q < n; ++q) {
            const int a = blockIdx.x + q * G;
            const bool on = mask == nullptr || mask[a] != 0;
            if (q >= K) lds_wait(c_stored, 4 * (q - K + 1));                // the buffer is free again
            if (on) {
                const TopBuf b = top_buf(p, bufs + (size_t)(q % K) * bw);
                top_prepare(p, a, b, tid);
                prepared += 1;
                lds_signal(c_prepared); lds_wait(c_prepared, 4 * prepared);  // planes cleared by all four wavefronts
#ifdef RCW_DEV_SWITCHES
                if (!(p.top_debug & 1))
#endif
                top_draw<T, TIE_LE, DIST_PRE>(p, a, b, tid);
            }
            lds_signal(c_drawn);
        }
    } else {
        for (int q = 0; q < n; ++q) {
            const int a = blockIdx.x + q * G;
            const bool on = mask == nullptr || mask[a] != 0;
            lds_wait(c_drawn, 4 * (q + 1));
#ifdef RCW_DEV_SWITCHES
            if (p.top_debug & 2) { lds_signal(c_stored); continue; }
#endif
            if (on) top_store(p, a, top_buf(p, bufs + (size_t)(q % K) * bw), tid);
            lds_signal(c_stored);
        }
    }
}

// ---- the top view as two kernels: draw (VALU/LDS work) | store (HBM work) ------------------------------------------
// A frame-per-workgroup store stream (the ring kernel above) tops out at ≈ 75 % of the HBM write peak on this chip;
// the camera fill's moving window — all wavefronts of the device sweeping ONE compact window of 1 KiB chunks —
// reaches 86 %.  The window needs every agent's line plane visible to every wavefront, so here the drawing is a
// kernel of its own that leaves the planes in HBM (Ht·Wt/8 bytes per agent, 1/32 of the image), and the store
// kernel is the fill kernel's sweep with the top view's pixel logic.  Inside a step the draw kernel runs on a side
// stream next to the camera fill (rcw_api.hip: launch_step) — one is VALU/LDS-bound, the other HBM-bound — so its
// time is hidden; stand-alone the two run back to back.
// Taken when a 1 KiB chunk (256 pixels of one image column) holds whole tiles and a lane's four pixels whole
// quarters of one: pu in {8, 16, 32, 64, 128, 256}, H·pu a multiple of 256; and the player's circle fits one
// 32-bit mask per image column (2·rp + 1 <= 32).  Other geometries keep the ring kernel.
//
// ---- EXPERIMENT, development build only (RCW_TOP_FOLLOW, docs/experiments.md): the store kernel FOLLOWS the draw kernel.  Launched
// on two streams with no event between them, the two run at once: the draw workgroup of agent a, when its plane, header and codes
// are in memory, adds one to the counter of the agent's BLOCK (2^p.top_blk_shift consecutive agents; the counters are never reset:
// after the call numbered p.top_epoch a complete block stands at epoch x its agents); a storing wavefront, before it loads anything
// of a group of 64 chunks, waits until every block up to the group's last agent is complete — 64 counters a look, one per lane.
// What the draw kernel publishes goes out as write-through stores (sc0 sc1: through the XCD's L2 to memory) — a release fence in
// front of the counter writes the WHOLE L2 back instead, the store kernel's gigabyte of pixels included, once per agent (100 us an
// agent); the counters are relaxed agent-scope atomics; a wavefront that has seen its blocks complete invalidates its caches once
// (acquire) and reads on with ordinary loads.  Bit-exact — and SLOWER than draw -> store back to back at every shape (a wavefront's
// look drains its stores, every advance invalidates an L2 under the window): rejected, not in the shipped library.
#ifdef RCW_DEV_SWITCHES
__device__ __forceinline__ void store_through(uint32_t* q, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(uint2* q, uint2 v)
{
    const unsigned long long w = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(q), "v"(w) : "memory");
}
__device__ __forceinline__ void top_publish(const RcwDev& p, int a)         // one thread of the draw workgroup, behind its last barrier
{
    if (p.top_debug & 16) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // (a full release in front of the counter)
    if (p.top_debug & 128) return;                                             // (nothing is published — the store kernel must give up)
    (void)__hip_atomic_fetch_add(p.top_flags + ((uint32_t)a >> p.top_blk_shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

#ifdef RCW_DEV_SWITCHES
// ---- the ROUND-4 draw body (development build only, RCW_TOP_DRAW=r4): for the comparison with top_draw_body below ----
// Draw kernel: one workgroup per agent; what the draw group of the ring kernel does, then the line plane is copied
// out unpadded, with the player's pixel (SR:468) and, per (tile column, row block), the 2-bit fill codes of the
// chunk's tiles packed into 64 bits.
template <typename T, bool TIE_LE, bool DIST_PRE>
__device__ __forceinline__ void top_draw_body_r4(const RcwDev& p, const uint8_t* __restrict__ mask, int a, uint32_t* lds)
{
    const int tid = threadIdx.x, group = blockDim.x;                          // p.top_draw_block threads: 256, or up to 1024 (a lane per ray) for big planes
    if (mask != nullptr && mask[a] == 0) return;                             // workgroup-uniform
    const TopBuf b = top_buf(p, lds);
    RCW_DRAW_STAMP(0);
    top_prepare(p, a, b, tid, group);
    __syncthreads();
    RCW_DRAW_STAMP(1);
    top_draw<T, TIE_LE, DIST_PRE>(p, a, b, tid, false, group);
    RCW_DRAW_STAMP(3);
    __syncthreads();
    RCW_DRAW_STAMP(4);
    const int pu = p.pu, Ht = p.H * pu, Wt = p.W * pu;
    if (tid == 0) p.top_hdr[a] = make_int2(b.hdr[0], b.hdr[1]);
    if (p.top_flat) {
        // rcw_top_store_flat_kernel's plane: the bit of agent pixel q = (j-1)·Ht + (i-1) sits at bit s + q of the agent's
        // region of p.top_plane_words words, s = (a · Ht·Wt) mod 256 — where the agent's image starts inside its first
        // 256-pixel chunk of the flat batch — so a chunk's plane bits are 8 whole words of the region, and the bits
        // that belong to the neighbouring agents' pixels (in front of s, behind the image) are zero: a chunk that
        // straddles two agents ORs the two regions' words.  An image column is at least 42 rows here, so a word holds
        // bits of at most two columns.
        const unsigned px_agent = (unsigned)Ht * (unsigned)Wt;
        const int s_a = (int)(((unsigned long long)a * px_agent) & 255ull);
        const unsigned cb = (unsigned)top_col_bits(p), PW = (unsigned)p.top_plane_words;
        uint32_t* const out = p.top_plane + (size_t)a * PW;
        for (unsigned w = tid; w < PW; w += group) {
            const int q_start = (int)(32u * w) - s_a;                        // the agent pixel of the word's bit 0
            uint32_t word = 0u;
            if (q_start > -32 && q_start < (int)px_agent) {
                const int lead = q_start < 0 ? -q_start : 0;
                const unsigned q = (unsigned)(q_start + lead);
                const unsigned j = q / (unsigned)Ht, i = q - j * (unsigned)Ht;
                const unsigned A = j * cb + i;
                const unsigned long long two = (unsigned long long)b.line[A >> 5] | ((unsigned long long)b.line[(A >> 5) + 1] << 32);
                uint32_t bits = (uint32_t)(two >> (A & 31u));
                const unsigned n1 = (unsigned)Ht - i;                        // bits left in column j
                if (n1 < 32u) {
                    bits &= (1u << n1) - 1u;
                    if (j + 1 < (unsigned)Wt) bits |= b.line[((j + 1) * cb) >> 5] << n1;
                }
                word = bits << lead;
            }
            out[w] = word;
        }
#ifdef RCW_TRACE_WAVES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RCW_DRAW_STAMP(5);
#endif
        return;
    }
    const int wpc = top_col_bits(p) >> 5, wpu = Ht >> 5, k = Ht / p.top_unit_px, tpc = p.top_unit_px / pu;   // (unit: 256 rows, or 128 / 64: rcw_top_store_units_kernel)
    uint32_t* const out = p.top_plane + (size_t)a * Wt * wpu;
    const int total = Wt * wpu, qstep = group / wpu, rstep = group - qstep * wpu;
    int j = tid / wpu, w = tid - j * wpu;
    for (int idx = tid; idx < total; idx += group) {
        out[idx] = b.line[j * wpc + w];
        j += qstep; w += rstep;
        if (w >= wpu) { w -= wpu; j += 1; }
    }
    for (int e = tid; e < p.W * k; e += group) {
        const int tj = e / k, rb = e - tj * k;
        const uint8_t* const tiles = b.tb + rb * tpc + p.H * tj;
        uint32_t lo = 0u, hi = 0u;
        for (int t = 0; t < tpc; ++t) {
            const uint32_t code = (tiles[t] & 1u) ? 1u : (tiles[t] & 2u);          // wall (white) before goal (red)  SR:355-360
            if (t < 16) lo |= code << (2 * t); else hi |= code << (2 * (t - 16));
        }
        p.top_codes[((size_t)a * p.W + tj) * k + rb] = make_uint2(lo, hi);
    }
#ifdef RCW_TRACE_WAVES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RCW_DRAW_STAMP(5);
    if (tid == 0 && a < 2048) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        g_draw_trace[a * 20 + 6] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
#endif
}

#endif   // RCW_DEV_SWITCHES (top_draw_body_r4)

// ---- the draw kernel's body (round 5) ------------------------------------------------------------------------------
// One workgroup per agent: rays -> lines in an LDS bit plane -> the plane (1/32 of the image) to HBM, with the player's pixel
// (SR:468) and, per (tile column, row block), the 2-bit fill codes of the chunk's tiles.  The kernel is bound by its INSTRUCTION
// COUNT (profiles/r05_draw_kernel.txt: 15 issue slots per pixel-step of a wavefront, and as many again per agent in set-up at
// cfg-2), so this body (a) asks for everything it needs from HBM in two batches, as rcw_cast_kernel does; (b) does not walk what
// another lane walks anyway — see top_covered_prefix: exact, the planes are bit for bit those of the round-4 body —; (c) has no
// integer or Float64 division in its set-up; (d) walks with a hand-scheduled loop of 8 vector instructions a pixel.
#ifdef RCW_DEV_SWITCHES
#define RCW_PLANE_STORE(q, v) do { if (p.top_signal) store_through((q), (v)); else *(q) = (v); } while (0)   // (the experiment above: write-through where the draw kernel publishes)
#else
#define RCW_PLANE_STORE(q, v) (*(q) = (v))
#endif
constexpr int kDrawRays = 2;                // rays a lane holds from the early table loads (more rays a lane take a loop)
constexpr uint32_t kNoLine = 0xFFFFFFFFu;   // the ray's line is not in the list: off-image end points (walked at once, clipped), or no such ray
// words of the draw kernel's LDS: header, tile bytes, line plane | the rays' end pixels [N] | what is left of each ray's line [N] | the lines to walk,
// sorted: end pixel [N], first pixel [N] | lines per length class [32] | where a class starts in the sorted list [32]
// (no circle plane here — the store kernels make the circle themselves — and no dummy words: lanes without a line aim at the class
// counters, which are dead by then.  At 768 x 768 px this is what lets TWO draw workgroups share a CU's 160 KiB: 79.8 KiB each.)
__host__ __device__ __forceinline__ size_t top_draw_lds_words(const RcwDev& p)
{
#ifdef RCW_DEV_SWITCHES
    if (p.top_draw_r4) return top_buf_words(p) + 4 * (size_t)((p.N + 3) & ~3) + 64;      // (the round-4 body: the one-kernel form's whole buffer in front)
#endif
    return 4 + top_tile_words(p) + top_line_words(p) + 4 * (size_t)((p.N + 3) & ~3) + 64;
}

// Wavefront-wide sums, maxima and prefix sums in the vector unit's data-parallel primitives (DPP: no LDS round trip, as __shfl takes
// through ds_bpermute): the sequences of AMD's cross-lane guide.  dpp0: the other lane's x, 0 where there is none or the row / bank is masked.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp0(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, BANK_MASK, true); }
__device__ __forceinline__ int wave_max_in_lane63(int x)                       // (x >= 0; quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, :8, row_bcast:15, :31)
{
    x = max(x, dpp0<0xB1>(x)); x = max(x, dpp0<0x4E>(x)); x = max(x, dpp0<0x124>(x)); x = max(x, dpp0<0x128>(x));
    x = max(x, dpp0<0x142, 0xA>(x)); x = max(x, dpp0<0x143, 0xC>(x));
    return x;
}
__device__ __forceinline__ int wave_prefix_sum(int x)                          // inclusive (row_shr:1, :2, :3, :4 banks 1-3, :8 banks 2-3, row_bcast:15, :31)
{
    int s = x + dpp0<0x111>(x);
    s += dpp0<0x112>(x);
    s += dpp0<0x113>(x);
    s += dpp0<0x114, 0xF, 0xE>(s);
    s += dpp0<0x118, 0xF, 0xC>(s);
    s += dpp0<0x142, 0xA>(s);
    s += dpp0<0x143, 0xC>(s);
    return s;
}

// floor(2^32 · b / a) for 0 <= b < a < 2^15 without a Float64 division: two 16-bit digits of the quotient, each a Float32 estimate
// repaired by fast_div's two corrections (its preconditions: n < 2^31 - d, quotient <= 2^16 at a relative error of ~2^-22)
__device__ __forceinline__ uint32_t line_slope(int lb, int la)
{
    const float inv = __builtin_amdgcn_rcpf((float)la);
    const int n1 = lb << 16;
    const int q1 = fast_div(n1, la, inv);
    const int n2 = (n1 - q1 * la) << 16;
    const int q2 = fast_div(n2, la, inv);
    return ((uint32_t)q1 << 16) + (uint32_t)q2;
}

// A line from the player's pixel (ip, jp) to (i2, j2) as SD.Line walks it (ASSUMED Bresenham, see top_draw): `a` steps along the
// major axis, pixel k at floor(k·b/a + 1/2) steps along the minor one; oct = which axis is major and the two step signs.
struct LineGeom { int a, b, oct; };
__device__ __forceinline__ LineGeom line_geom(uint32_t key, int ip, int jp)
{
    const int i2 = (int)(key & 0xFFFFu), j2 = (int)(key >> 16);
    const int di = abs(i2 - ip), dj = abs(j2 - jp);
    LineGeom g;
    const bool imaj = di >= dj;
    g.a = imaj ? di : dj; g.b = imaj ? dj : di;
    g.oct = (imaj ? 1 : 0) | (ip < i2 ? 2 : 0) | (jp < j2 ? 4 : 0);
    return g;
}
// How many leading pixels k = 0 .. K-1 of the line `mine` need not be drawn because the lines `lo` and `hi` — the rays 2^t before
// and behind it in the fan — draw them: all three start at the player's pixel; in the same octant pixel k of each sits k steps
// along the major axis and floor(k·s + 1/2) along the minor one, s = b/a; with s_lo <= s_mine <= s_hi (or the reverse) the middle
// line's pixel lies between the outer ones, and while k·|s_hi - s_lo| < 1 those are at most one apart: it IS one of them.  All in
// exact integers (a, b < 2^14: the cross products fit 32 bits) but the last division, whose Float32 estimate is taken low (fewer
// pixels skipped, never one too many).  Identical end points: the whole line (K = a + 1).  tests/test_host_logic.py replays this
// against the union of all lines.
__device__ __forceinline__ int top_covered_prefix(uint32_t key, const LineGeom& m, uint32_t key_lo, uint32_t key_hi, int ip, int jp)
{
    if (key_lo == kNoLine || key_hi == kNoLine) return 0;
    if (key == key_lo || key == key_hi) return m.a + 1;
    const LineGeom l = line_geom(key_lo, ip, jp), h = line_geom(key_hi, ip, jp);
    if (l.oct != m.oct || h.oct != m.oct) return 0;
    // (extents below 2^14: v_mul_i32_i24 — full rate — gives the whole product)
    const int lm = __mul24(l.b, m.a) - __mul24(m.b, l.a), mh = __mul24(m.b, h.a) - __mul24(h.b, m.a);   // s_l - s_m and s_m - s_h, scaled by positive numbers
    if (!((lm <= 0 && mh <= 0) || (lm >= 0 && mh >= 0))) return 0;          // not monotone
    const int P = abs(__mul24(l.b, h.a) - __mul24(h.b, l.a)), Q = __mul24(l.a, h.a);   // |s_l - s_h| = P / Q
    int kmax = l.a < h.a ? l.a : h.a;                                        // both outer lines have a pixel k only up to their own length
    if (P > 0) {
        const int kstar = (int)((float)(Q - 1) * __builtin_amdgcn_rcpf((float)P) * 0.99999f) - 1;   // < Q / P, taken low
        kmax = kstar < kmax ? kstar : kmax;
    }
    kmax = kmax < m.a ? kmax : m.a;
    return kmax < 0 ? 0 : kmax + 1;                                          // pixels 0 .. kmax
}

// SD.Line with an end point off the image (SimpleDraw skips the pixels off it): the error-term loop as written
__device__ __forceinline__ void top_clipped_line(uint32_t* line, int cb_, int Ht, int Wt, int ip, int jp, int i2, int j2)
{
    int i1 = ip, j1 = jp;
    const int di = abs(i2 - i1), dj = -abs(j2 - j1);
    const int si = i1 < i2 ? 1 : -1, sj = j1 < j2 ? 1 : -1;
    int err = di + dj;
    for (long long guard = 0; guard <= (long long)di - dj; ++guard) {
        if (i1 >= 1 && i1 <= Ht && j1 >= 1 && j1 <= Wt) {
            const int q = (j1 - 1) * cb_ + (i1 - 1);
            __hip_atomic_fetch_or(line + (q >> 5), 1u << (q & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (i1 == i2 && j1 == j2) break;
        const int e2 = 2 * err;
        if (e2 >= dj) { err += dj; i1 += si; }
        if (e2 <= di) { err += di; j1 += sj; }
    }
}

__device__ __forceinline__ uint32_t lds_address(const void* q) { return (uint32_t)reinterpret_cast<size_t>((__attribute__((address_space(3))) const uint8_t*)q); }

// One pixel-step of the walk for every lane, one asm statement: OR the pixel's bit into the plane, advance along the line, count the
// steps to the end of the lane's segment down — and, where some lane's segment just ended (vcc, rare against the steps: once per lane
// and time round its segment), send those lanes back to its start.  A is the BIT address of the pixel in LDS (the plane's own
// address folded in), f the 32-bit fraction whose carry steps the minor axis (see top_draw).  Hand-scheduled: a VALU instruction that
// reads vcc needs two others between it and the one that wrote it (gfx950); 8 vector instructions, the ds_or and one branch a pixel
// (the compiler's loop of round 4: 9 + 1 + a nop + 4 scalar).
#define RCW_DRAW_STEP(A, f, rem, slope, smaj, sboth, t, m, dd, rem0, f0, A_0)                                               \
    asm volatile("v_add_co_u32_e32 %1, vcc, %1, %6\n\t"                                                                     \
                 "v_lshrrev_b32_e32 %3, 3, %0\n\t"                                                                           \
                 "v_lshlrev_b32_e64 %4, %0, 1\n\t"                                                                           \
                 "v_cndmask_b32_e32 %5, %7, %8, vcc\n\t"                                                                     \
                 "v_and_b32_e32 %3, 0x1ffffffc, %3\n\t"                                                                      \
                 "v_subrev_co_u32_e32 %2, vcc, 1, %2\n\t"                                                                    \
                 "ds_or_b32 %3, %4\n\t"                                                                                      \
                 "v_add_u32_e32 %0, %0, %5\n\t"                                                                              \
                 "s_cbranch_vccz 1f\n\t"                                                                                     \
                 "v_cndmask_b32_e32 %2, %2, %9, vcc\n\t"                                                                     \
                 "v_cndmask_b32_e32 %1, %1, %10, vcc\n\t"                                                                    \
                 "v_cndmask_b32_e32 %0, %0, %11, vcc\n"                                                                      \
                 "1:"                                                                                                        \
                 : "+v"(A), "+v"(f), "+v"(rem), "=&v"(t), "=&v"(m), "=&v"(dd)                                                \
                 : "v"(slope), "v"(smaj), "v"(sboth), "v"(rem0), "v"(f0), "v"(A_0) : "vcc", "memory")

#ifdef RCW_DEV_SWITCHES   // (timing probe RCW_TOP_DRAW=halfds: the same step without its LDS atomic — wrong pixels, see profiles/r05_draw_kernel.txt)
#define RCW_DRAW_STEP_NO_LDS(A, f, rem, slope, smaj, sboth, t, m, dd, rem0, f0, A_0)                                               \
    asm volatile("v_add_co_u32_e32 %1, vcc, %1, %6\n\t"                                                                     \
                 "v_lshrrev_b32_e32 %3, 3, %0\n\t"                                                                           \
                 "v_lshlrev_b32_e64 %4, %0, 1\n\t"                                                                           \
                 "v_cndmask_b32_e32 %5, %7, %8, vcc\n\t"                                                                     \
                 "v_and_b32_e32 %3, 0x1ffffffc, %3\n\t"                                                                      \
                 "v_subrev_co_u32_e32 %2, vcc, 1, %2\n\t"                                                                    \
                 "v_add_u32_e32 %0, %0, %5\n\t"                                                                              \
                 "s_cbranch_vccz 1f\n\t"                                                                                     \
                 "v_cndmask_b32_e32 %2, %2, %9, vcc\n\t"                                                                     \
                 "v_cndmask_b32_e32 %1, %1, %10, vcc\n\t"                                                                    \
                 "v_cndmask_b32_e32 %0, %0, %11, vcc\n"                                                                      \
                 "1:"                                                                                                        \
                 : "+v"(A), "+v"(f), "+v"(rem), "=&v"(t), "=&v"(m), "=&v"(dd)                                                \
                 : "v"(slope), "v"(smaj), "v"(sboth), "v"(rem0), "v"(f0), "v"(A_0) : "vcc", "memory")
#endif

// LDS atomic add that returns the old value, by name: through __hip_atomic_fetch_add the compiler wraps every such add in a
// wavefront-wide reduction loop (its atomic optimizer), two dozen instructions where one is meant
__device__ __forceinline__ uint32_t lds_add_return(uint32_t* counter, uint32_t v)
{
    uint32_t old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(lds_address(counter)), "v"(v) : "memory");
    return old;
}
constexpr int kDrawBuckets = 32;            // the lines to walk are sorted by length into this many classes, longest first

template <typename T, bool TIE_LE, bool DIST_PRE>
__device__ __forceinline__ void top_draw_body(const RcwDev& p, const uint8_t* __restrict__ mask, int a, uint32_t* lds, int part = 0, int parts = 1)
{
    // (parts > 1: this workgroup walks the lines of rays [ray_lo, ray_hi) of the agent's fan only — the other parts' workgroups, on
    // other CUs, the rest — and ORs its plane into the agent's plane in HBM, which the store kernel leaves zeroed: rcw_top_draw_kernel)
    typedef typename Real<T>::vec2 vec2;
    const int tid = threadIdx.x, group = blockDim.x, lane = tid & 63;        // p.top_draw_block threads: 256, or up to 1024 (a lane per ray) for big planes
    const int H = p.H, HW = p.H * p.W, N = p.N, pu = p.pu, Ht = H * pu, Wt = p.W * pu;
    const TopBuf b = top_buf(p, lds);
    const int ray_lo = (int)((long long)N * part / parts), ray_hi = (int)((long long)N * (part + 1) / parts);
    const int npad4 = (N + 3) & ~3;
    uint32_t* const ends = b.line + top_line_words(p);                       // [N] the rays' end pixels (i2 | j2 << 16), kNoLine: none
    uint32_t* const meta = ends + npad4;                                     // [N] per ray: pixels left out | length class << 15 | rank in the class << 20
    uint32_t* const sorted_key = meta + npad4;                               // [M] the lines to walk, longest first: end pixel
    uint32_t* const sorted_first = sorted_key + npad4;                       // [M] ... and the first pixel to walk
    uint32_t* const bcount = sorted_first + npad4;                           // [kDrawBuckets] lines per length class
    volatile uint32_t* const bstart = bcount + kDrawBuckets;                 // [kDrawBuckets] ... and where the class starts in the sorted list
    RCW_DRAW_STAMP(0);
#ifdef RCW_TRACE_WAVES
    if (tid == 0 && a < 2048) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        g_draw_trace[a * 20 + 6] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
#endif

    // ---- batch 1: the agent's state (mask byte, heading, pose as scalar loads awaited once: load_cast_state), the lane's tile-map words
    const uint32_t* const tm_hbm = p.tile_map + (size_t)a * p.nwords;
    uint32_t tw[kCastTiles];
#pragma unroll
    for (int k = 0; k < kCastTiles; ++k) {
        const int t = tid + k * group;
        tw[k] = load_at(tm_hbm, (uint32_t)((t < HW ? t : HW - 1) >> 4) * 4u);
    }
    const uint8_t* const mask_q = mask != nullptr ? mask + a : p.done + a;
    vec2 pos;
    const CastState st = load_cast_state(mask_q, p.done + a, p.done + a, p.dir + a, Real<T>::pos(p) + a, pos);
    if (mask != nullptr && byte_of_word(st.mask_w, mask_q) == 0) return;     // workgroup-uniform
    // ---- batch 2: the heading's ray-table entries of this lane's first rays, in flight while LDS is set up
    const T* const tab = Real<T>::ray_table(p) + (size_t)st.d * RCW_TABLE_ROWS * N;
    T r_dx[kDrawRays], r_dy[kDrawRays], r_ddx[kDrawRays], r_ddy[kDrawRays];
#pragma unroll
    for (int k = 0; k < kDrawRays; ++k) {
        const int i = tid + k * group;
        const uint32_t o = (uint32_t)(i < N ? i : N - 1) * (uint32_t)sizeof(T);   // (lanes past the last ray re-read it)
        r_dx[k] = load_at(tab, o); r_dy[k] = load_at(tab + N, o); r_ddx[k] = load_at(tab + 2 * N, o); r_ddy[k] = load_at(tab + 3 * N, o);
    }
    // ---- LDS: the tile bytes (the last tile an obstacle whatever HBM holds: stage_tile_bytes), the cleared line plane, the class counts
#pragma unroll
    for (int k = 0; k < kCastTiles; ++k) {
        const int t = tid + k * group;
        if (t < HW) { const uint32_t v = (tw[k] >> ((t & 15) * 2)) & 3u; b.tb[t] = (uint8_t)(t == HW - 1 ? (v | 1u) : v); }
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int t = tid + kCastTiles * group; t < HW; t += group) {
        const uint32_t v = (tm_hbm[t >> 4] >> ((t & 15) * 2)) & 3u;
        b.tb[t] = (uint8_t)(t == HW - 1 ? (v | 1u) : v);
    }
    {
        u32x4* const z = reinterpret_cast<u32x4*>(b.line);
        const int nz = (int)(top_line_words(p) >> 2);
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int k = tid; k < nz; k += group) z[k] = zero;
    }
    if (tid < kDrawBuckets) bcount[tid] = 0u;
    __syncthreads();
    RCW_DRAW_STAMP(1);

    // ---- the rays' end pixels (SR:476) ----------------------------------------------------------------------------------
    const int ip = wu_to_pu<T>(pos.x, pu), jp = wu_to_pu<T>(pos.y, pu);      // SR:468 (1-based)
    const bool start_inside = ip >= 1 && ip <= Ht && jp >= 1 && jp <= Wt;
    const int cb_ = top_col_bits(p);
    auto end_pixel = [&](int i, T dx, T dy, T ddx, T ddy) {
        const RayHit<T> r = cast_ray<T, TIE_LE, DIST_PRE>(b.tb, p.H, p.W, pos.x, pos.y, dx, dy, ddx, ddy);
        const T dist = r.oob ? (T)0 : r.dist;
        const T ox = dist * dx, oy = dist * dy;                              // ray_distance_wu * ray_direction_wu
        const T ex = pos.x + ox, ey = pos.y + oy;
        const int i2 = wu_to_pu<T>(ex, pu), j2 = wu_to_pu<T>(ey, pu);       // SR:476
        // a line whose end points are both on the image stays on it; anything else is walked here and now, clipped
        const bool inside = start_inside && i2 >= 1 && i2 <= Ht && j2 >= 1 && j2 <= Wt;
        ends[i] = inside ? (uint32_t)i2 | ((uint32_t)j2 << 16) : kNoLine;
        if (!inside && i >= ray_lo && i < ray_hi) top_clipped_line(b.line, cb_, Ht, Wt, ip, jp, i2, j2);
    };
#pragma unroll
    for (int k = 0; k < kDrawRays; ++k) {
        const int i = tid + k * group;
        if (i < N) end_pixel(i, r_dx[k], r_dy[k], r_ddx[k], r_ddy[k]);
    }
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int i = tid + kDrawRays * group; i < N; i += group) end_pixel(i, tab[i], tab[N + i], tab[2 * N + i], tab[3 * N + i]);
    __syncthreads();

    // ---- which pixels of which lines have to be walked: ray r = 2^t (2 m + 1) leaves to the rays r - 2^t and r + 2^t what they
    // draw of its line (top_covered_prefix; their own omissions are drawn by rays of still higher t: no cycle); ray 0 and rays
    // without both such neighbours walk everything.  What is left is SORTED by length (a counting sort over kDrawBuckets classes of
    // the image's longer side, longest first): the walk below is as long as a wavefront's longest line, and the remainders differ
    // a lot — half the rays keep a fraction of their line or nothing, a few keep all of it.
    int len_shift = 0;
    while (((Ht > Wt ? Ht : Wt) >> len_shift) >= kDrawBuckets) ++len_shift;  // (a line has at most max(Ht, Wt) pixels)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int i = tid; i < N; i += group) {
        const uint32_t key = ends[i];
        if (key == kNoLine) continue;
        if (i < ray_lo || i >= ray_hi) { meta[i] = 0x7FFFu; continue; }      // another part's line
        const LineGeom g = line_geom(key, ip, jp);
        int skip = 0;
        if (i > 0) {
            const int t = i & -i;                                            // 2^t, t = the trailing zeros of i (i - 2^t >= 0 by construction)
            if (i + t < N) skip = top_covered_prefix(key, g, ends[i - t], ends[i + t], ip, jp);
        }
        const int n = g.a + 1 - skip;                                        // pixels skip .. a to walk
        uint32_t m = 0x7FFFu;                                                // (nothing of this line is walked)
        if (n > 0) {
            int cls = kDrawBuckets - 1 - min(kDrawBuckets - 1, (n - 1) >> len_shift);
#ifdef RCW_DEV_SWITCHES
            if (p.top_draw_banks == 1) {                                     // (experiment) four kinds of line x eight classes of length: a wavefront's lanes then move through the banks alike
                const int kind = (g.oct & 1) | ((((g.oct & 1) ? (g.oct >> 1) : (g.oct >> 2)) & 1) << 1);
                cls = kind * 8 + 7 - min(7, (n - 1) >> (len_shift + 2));
            }
#endif
            m = (uint32_t)skip | ((uint32_t)cls << 15) | (lds_add_return(bcount + cls, 1u) << 20);
        }
        meta[i] = m;
    }
    __syncthreads();
    int M;
    {
        const int mine = lane < kDrawBuckets ? (int)bcount[lane] : 0;
        const int incl = wave_prefix_sum(mine);
        M = __builtin_amdgcn_readlane(incl, 63);                             // the lines to walk
        // lane c: the lines of longer classes = where class c starts in the sorted list.  Every wavefront writes the same 32 words and
        // reads them back itself (a wavefront's LDS operations execute in order: no barrier)
        if (lane < kDrawBuckets) bstart[lane] = (uint32_t)(incl - mine);
        __builtin_amdgcn_wave_barrier();
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int i = tid; i < N; i += group) {
            const uint32_t key = ends[i];
            if (key == kNoLine) continue;
            const uint32_t m = meta[i];
            if ((m & 0x7FFFu) == 0x7FFFu) continue;
            const uint32_t at = bstart[(m >> 15) & 31u] + (m >> 20);
            sorted_key[at] = key; sorted_first[at] = m & 0x7FFFu;
        }
    }
    __syncthreads();
    RCW_DRAW_STAMP(2);

    // ---- the walk: a lane per line, or — fewer lines than lanes — 2^lp lanes per line, a segment each; in passes of the workgroup ----
    // Pixel k of a line sits k steps along the major axis and floor(k·b/a + 1/2) along the minor one.  The loop carries the
    // FRACTION of k·slope/2^32 + 1/2 + 2^-18 in 32 bits and steps the minor axis on its carry (exact for lines of up to 2^14
    // pixels: the argument is in top_draw).  Every lane walks its whole segment but starts somewhere along it and wraps round:
    // walked in step from the player, the lanes of a wavefront sit on one small arc at every step — the same plane word or two
    // for dozens of steps, and same-word LDS atomics serialise.  A lane that is through before the wavefront's longest segment
    // goes round again (OR is idempotent); a lane without a segment ORs into a private dummy word.
    if (M > 0) {
        int lp = 0;
        { const int mpad = (M + 63) & ~63, g64 = group >> 6; while ((mpad << (lp + 1)) <= group && (g64 & ((2 << lp) - 1)) == 0) ++lp; }
        const int rpp = group >> lp;                                         // lines per pass, a multiple of 64: the part is wave-uniform
        int seg = 0;
        { const int wpp = rpp >> 6, wv = tid >> 6; for (int t = wpp; t <= wv; t += wpp) ++seg; }
        const int pin = tid - seg * rpp;
        const uint32_t plane_bits = lds_address(b.line) * 8u;
        const uint32_t dummy_A = lds_address(bcount + lane) * 8u;              // (the 64 words of the class counters and starts: used up by now)
        const uint32_t frac0 = 0x80000000u + (1u << 14);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int q0 = 0; q0 < M; q0 += rpp) {
            const int q = q0 + pin;
            int ks = 0, len = 0, smaj = 0, sboth = 0;
            uint32_t A0 = dummy_A, slope = 0u;
            if (q < M) {
                const uint32_t key = sorted_key[q];
                const int first = (int)sorted_first[q];
                const LineGeom g = line_geom(key, ip, jp);
                const int n = g.a + 1 - first;
                const int si = (g.oct & 2) ? 1 : -1, sj = (g.oct & 4) ? cb_ : -cb_;   // steps of the plane's bit index
                smaj = (g.oct & 1) ? si : sj;
                sboth = si + sj;
                A0 = plane_bits + (uint32_t)((jp - 1) * cb_ + (ip - 1));
                slope = g.b >= g.a ? 0xFFFFFFFFu : line_slope(g.b, g.a);      // floor(2^32 · b / a)
                ks = first + ((seg * n) >> lp);                             // this lane's pixels of the line: ks .. ke - 1
                len = first + (((seg + 1) * n) >> lp) - ks;
                if (len == 0) { A0 = dummy_A; smaj = sboth = 0; slope = 0u; }
            }
            const int ke = ks + len;
            int k0 = len > 1 ? ks + (int)((((unsigned)(tid * 37) & 63u) * (unsigned)len) >> 6) : ks;   // neighbouring lanes start 37/64 of a segment apart
#ifdef RCW_DEV_SWITCHES
            if (p.top_draw_banks == 1 && len > 48) {
                // (experiment) move the start on by up to 31 bank steps so that lane l of a half-wavefront starts on bank l: a step along a
                // column-major line, or a minor step of a row-major one, moves the word by the plane's (odd) column stride
                const unsigned long long at0 = (unsigned long long)(unsigned)k0 * slope + frac0;
                const uint32_t Ab = A0 + (uint32_t)(k0 * smaj + (int)(at0 >> 32) * (sboth - smaj));
                const int w = cb_ >> 5, inv_w = (w * (2 - w * w)) & 31;    // the stride's inverse modulo 32 (w odd)
                const int col_step = (abs(smaj) > 1) ? smaj : (sboth - smaj);                  // the one of the two steps that changes the column
                int turns = ((((lane & 31) - (int)((Ab >> 5) & 31u)) & 31) * inv_w) & 31;      // column steps forward to the wanted bank ...
                if (col_step < 0) turns = (32 - turns) & 31;                                    // ... or as many backward
                int kn = k0;
                if (abs(smaj) > 1) kn = k0 + turns;                                             // column-major line: a column a step
                else if (slope != 0u) kn = k0 + (int)((float)turns * 4294967296.0f / (float)slope);   // row-major line: a column every 2^32 / slope steps (roughly: any start is a valid start)
                if (kn >= ke) kn -= 32;
                if (kn >= ks && kn < ke) k0 = kn;
            }
#endif
            const unsigned long long at_ks = (unsigned long long)(unsigned)ks * slope + frac0;     // v_mad_u64_u32
            const unsigned long long at_k0 = (unsigned long long)(unsigned)k0 * slope + frac0;
            const uint32_t frac_s = (uint32_t)at_ks;
            const uint32_t A_s = A0 + (uint32_t)(ks * smaj + (int)(at_ks >> 32) * (sboth - smaj));
            uint32_t frac = (uint32_t)at_k0;
            uint32_t A = A0 + (uint32_t)(k0 * smaj + (int)(at_k0 >> 32) * (sboth - smaj));
            uint32_t rem = (uint32_t)(len > 0 ? ke - k0 : 0x7fffffff) - 1u;   // steps until the wrap, less one (the step that borrows wraps)
            const uint32_t len_m1 = len > 0 ? (uint32_t)(len - 1) : 0x7ffffffeu;
            const int nmax = __builtin_amdgcn_readlane(wave_max_in_lane63(len), 63);   // (the trip count is the wavefront's longest segment: a scalar loop)
#ifdef RCW_TRACE_WAVES
            if (q0 == 0 && tid == 0 && a < 2048) g_draw_trace[a * 20 + 8] = (unsigned long long)nmax | ((unsigned long long)M << 32);
#endif
            uint32_t t_, m_, d_;
            // four steps a trip (up to three more than the longest segment needs: lanes go round their own segments, harmless)
#ifdef RCW_DEV_SWITCHES
            if (p.top_draw_banks == 2) {                                     // (timing probe: every other step without its LDS atomic)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int trips = (nmax + 3) >> 2; trips > 0; --trips) {
                    RCW_DRAW_STEP(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                    RCW_DRAW_STEP_NO_LDS(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                    RCW_DRAW_STEP(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                    RCW_DRAW_STEP_NO_LDS(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                }
            } else if (p.top_draw_banks == 3) {                              // (timing probe: no step with an LDS atomic)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int trips = (nmax + 3) >> 2; trips > 0; --trips) {
                    RCW_DRAW_STEP_NO_LDS(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                    RCW_DRAW_STEP_NO_LDS(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                    RCW_DRAW_STEP_NO_LDS(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                    RCW_DRAW_STEP_NO_LDS(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                }
            } else
#endif
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (int trips = (nmax + 3) >> 2; trips > 0; --trips) {
                RCW_DRAW_STEP(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                RCW_DRAW_STEP(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                RCW_DRAW_STEP(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
                RCW_DRAW_STEP(A, frac, rem, slope, smaj, sboth, t_, m_, d_, len_m1, frac_s, A_s);
            }
        }
    }
    RCW_DRAW_STAMP(3);
    // (the walk's ds_or_b32 sit in asm statements, which the compiler's wait-count pass does not see: without the explicit
    // lgkmcnt(0) a wavefront could pass the barrier with plane ORs still in flight while others read b.line[] below)
    lds_barrier();
    RCW_DRAW_STAMP(4);
    if (tid == 0 && part == 0) RCW_PLANE_STORE(reinterpret_cast<uint2*>(p.top_hdr + a), make_uint2((uint32_t)ip, (uint32_t)jp));
    if (p.top_flat) {
        // rcw_top_store_flat_kernel's plane: the bit of agent pixel q = (j-1)·Ht + (i-1) sits at bit s + q of the agent's
        // region of p.top_plane_words words, s = (a · Ht·Wt) mod 256 — where the agent's image starts inside its first
        // 256-pixel chunk of the flat batch — so a chunk's plane bits are 8 whole words of the region, and the bits
        // that belong to the neighbouring agents' pixels (in front of s, behind the image) are zero: a chunk that
        // straddles two agents ORs the two regions' words.  An image column is at least 42 rows here, so a word holds
        // bits of at most two columns.
        const unsigned px_agent = (unsigned)Ht * (unsigned)Wt;
        const int s_a = (int)(((unsigned long long)a * px_agent) & 255ull);
        const unsigned cb = (unsigned)cb_, PW = (unsigned)p.top_plane_words;
        const float inv_ht = 1.0f / (float)Ht;
        uint32_t* const out = p.top_plane + (size_t)a * PW;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (unsigned w = tid; w < PW; w += group) {
            const int q_start = (int)(32u * w) - s_a;                        // the agent pixel of the word's bit 0
            uint32_t word = 0u;
            if (q_start > -32 && q_start < (int)px_agent) {
                const int lead = q_start < 0 ? -q_start : 0;
                const unsigned q = (unsigned)(q_start + lead);
                // (the image column of pixel q: the Float32 quotient repaired, where fast_div's precondition holds — q < 2^23, or a
                // quotient of at most 2^13; every image the flat store kernel takes is at most 2^14 pixels wide: the second holds)
                const unsigned j = (unsigned)fast_div((int)q, Ht, inv_ht), i = q - j * (unsigned)Ht;
                const unsigned Ab = j * cb + i;
                const unsigned long long two = (unsigned long long)b.line[Ab >> 5] | ((unsigned long long)b.line[(Ab >> 5) + 1] << 32);
                uint32_t bits = (uint32_t)(two >> (Ab & 31u));
                const unsigned n1 = (unsigned)Ht - i;                        // bits left in column j
                if (n1 < 32u) {
                    bits &= (1u << n1) - 1u;
                    if (j + 1 < (unsigned)Wt) bits |= b.line[((j + 1) * cb) >> 5] << n1;
                }
                word = bits << lead;
            }
            RCW_PLANE_STORE(out + w, word);
        }
#ifdef RCW_TRACE_WAVES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RCW_DRAW_STAMP(5);
#endif
#ifdef RCW_DEV_SWITCHES
    if (p.top_signal) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (tid == 0) top_publish(p, a); }   // (the write-through stores above: written by name, awaited by name)
#endif
        return;
    }
    const int wpc = cb_ >> 5, wpu = Ht >> 5, k = Ht / p.top_unit_px, tpc = p.top_unit_px / pu;   // (unit: 256 rows, or 128 / 64: rcw_top_store_units_kernel)
    uint32_t* const out = p.top_plane + (size_t)a * Wt * wpu;
    const int total = Wt * wpu, qstep = group / wpu, rstep = group - qstep * wpu;
    int j = tid / wpu, w = tid - j * wpu;
    if (parts > 1) {
        // several workgroups an agent: every one ORs the words it has bits in into the agent's plane (relaxed atomics without a
        // return value; the store kernel has left the plane zero: top_group_issue)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int idx = tid; idx < total; idx += group) {
            const uint32_t word = b.line[j * wpc + w];
            if (word != 0u) (void)__hip_atomic_fetch_or(out + idx, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            j += qstep; w += rstep;
            if (w >= wpu) { w -= wpu; j += 1; }
        }
        if (part != 0) return;                                               // (the tile codes: the first part's)
    } else if ((wpu & 3) == 0 && !p.top_signal) {
        // four words of a column a thread, one 16-byte store (a column is a multiple of 8 words here; in LDS its stride is odd: four 4-byte reads)
        const int qpc = wpu >> 2, quads = Wt * qpc, qs = group / qpc, rs = group - qs * qpc;
        int jq = tid / qpc, wq = tid - jq * qpc;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int q = tid; q < quads; q += group) {
            const uint32_t* const src = b.line + jq * wpc + 4 * wq;
            const u32x4 v = {src[0], src[1], src[2], src[3]};
            *reinterpret_cast<u32x4*>(out + jq * wpu + 4 * wq) = v;
            jq += qs; wq += rs;
            if (wq >= qpc) { wq -= qpc; jq += 1; }
        }
    } else
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int idx = tid; idx < total; idx += group) {
        RCW_PLANE_STORE(out + idx, b.line[j * wpc + w]);
        j += qstep; w += rstep;
        if (w >= wpu) { w -= wpu; j += 1; }
    }
    for (int e = tid; e < p.W * k; e += group) {
        const int tj = e / k, rb = e - tj * k;
        const uint8_t* const tiles = b.tb + rb * tpc + p.H * tj;
        uint32_t lo = 0u, hi = 0u;
        for (int t = 0; t < tpc; ++t) {
            const uint32_t code = (tiles[t] & 1u) ? 1u : (tiles[t] & 2u);          // wall (white) before goal (red)  SR:355-360
            if (t < 16) lo |= code << (2 * t); else hi |= code << (2 * (t - 16));
        }
        RCW_PLANE_STORE(p.top_codes + ((size_t)a * p.W + tj) * k + rb, make_uint2(lo, hi));
    }
#ifdef RCW_TRACE_WAVES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RCW_DRAW_STAMP(5);
#endif
#ifdef RCW_DEV_SWITCHES
    if (p.top_signal) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (tid == 0) top_publish(p, a); }   // (the write-through stores above: written by name, awaited by name)
#endif
}

template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(1024) void rcw_top_draw_kernel(const RcwDev p, const uint8_t* __restrict__ mask, int first)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
#ifdef RCW_DEV_SWITCHES
    if (p.top_draw_r4) { top_draw_body_r4<T, TIE_LE, DIST_PRE>(p, mask, first + (int)blockIdx.x, lds); return; }
#endif
    // p.top_parts workgroups an agent (1, or 2 .. 4 where a batch of big images leaves CUs without a workgroup, or one agent has a CU to
    // itself and the slowest agent is the kernel): workgroup q draws part q mod parts of agent q / parts — neighbours in the dispatch
    // order, i.e. on different XCDs
    const int parts = p.top_parts > 1 ? p.top_parts : 1;
    if (parts == 1) { top_draw_body<T, TIE_LE, DIST_PRE>(p, mask, first + (int)blockIdx.x, lds); return; }
    top_draw_body<T, TIE_LE, DIST_PRE>(p, mask, first + (int)blockIdx.x / parts, lds, (int)blockIdx.x % parts, parts);
}

// The camera fill and the top view's drawing in ONE launch (a step that renders both images, H_cam = 256, planes that fit a
// 256-thread draw workgroup): workgroups 0 .. fill_blocks - 1 are rcw_fill256_kernel's — dispatched first, onto an empty device,
// one per CU as in a launch of their own —, workgroup fill_blocks + q draws agent first + q.  What the side stream gave — an
// HBM-bound kernel and a VALU/LDS-bound one sharing the CUs — without its event record / wait pairs on two streams (DESIGN.md
// §4.4: ≈ 7–10 µs a step, and the reason small batches stayed on the one-kernel form).  The fill workgroups reserve the draw's
// LDS (one workgroup per CU: nothing else wanted it) and run at the draw's register count (they are one wavefront per SIMD).
template <typename T, bool TIE_LE, bool DIST_PRE>
__global__ __launch_bounds__(kBlock) void rcw_fill256_draw_kernel(const RcwDev p, const int32_t* __restrict__ col_h,
                                                                  const uint8_t* __restrict__ col_c, u32x4* __restrict__ out,
                                                                  long long total_cols, const uint8_t* __restrict__ mask,
                                                                  int fill_blocks, int first)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if ((int)blockIdx.x < fill_blocks) { fill256_body<false>(p, col_h, col_c, out, total_cols, mask, (int)blockIdx.x, fill_blocks); return; }
#ifdef RCW_DEV_SWITCHES
    if (p.top_draw_r4) { top_draw_body_r4<T, TIE_LE, DIST_PRE>(p, mask, first + (int)blockIdx.x - fill_blocks, lds); return; }
#endif
    top_draw_body<T, TIE_LE, DIST_PRE>(p, mask, first + (int)blockIdx.x - fill_blocks, lds);
}

// Store kernel: the moving window of rcw_fill256_kernel over the top view's 1 KiB chunks (chunk id = flat pixel
// offset / 256: image column (a, j), row block rb).  Per 64 chunks of a wavefront, lane l computes the descriptor
// of the l-th (tile codes, frame column, the player's circle as a 32-bit row mask for that column), and the 8 plane
// words of each chunk are fetched by 8 lanes (8 loads per lane for the 64 chunks); per chunk the descriptor is
// broadcast with v_readlane, the plane words go through a wave-private 2 KiB of LDS, and lane l writes rows 4l..4l+3
// with one 16-byte store: colour = circle > ray line > tile frame > tile fill (SR:362-367, SR:473-477, SR:480).
// The circle: SD.Circle's pixels in the image column at distance c from the player's are the same rows relative to
// the player for every agent (midpoint circle, assumed): lane c computes that row mask once per kernel.
// what depends on the lane only
struct TopLane {
    int r_lane, sh, code_sh;
    bool code_hi, first_row, last_row;
    uint32_t cm;                 // lane c: the circle's rows at column distance c
};
__device__ __forceinline__ TopLane top_lane(const RcwDev& p, int lane)
{
    TopLane L;
    const int pu = p.pu, rp = p.top_rp;
    L.r_lane = lane * 4;
    const int tl = L.r_lane / pu, ri = L.r_lane - tl * pu;                   // tile within the chunk, row within the tile
    L.first_row = ri == 0; L.last_row = ri + 3 == pu - 1;                    // SR:364-365: the tile's frame rows
    L.sh = L.r_lane & 31;
    L.code_sh = 2 * (tl & 15);
    L.code_hi = tl >= 16;
    L.cm = 0u;
    int x = 0, y = rp, dd = 1 - rp;
    while (x <= y) {
        if (y == lane) L.cm |= (1u << (rp + x)) | (1u << (rp - x));
        if (x == lane) L.cm |= (1u << (rp + y)) | (1u << (rp - y));
        x += 1;
        if (dd < 0) dd += 2 * x + 1;
        else { y -= 1; dd += 2 * (x - y) + 1; }
    }
    return L;
}

// v_bfe_i32 (one bit, sign-extended: 0 or ~0) and v_bfi_b32 by name: written in C the compiler turns the pair into
// and + compare + select, three instructions a pixel instead of two — and this kernel's wavefronts (one per SIMD, as
// the moving window wants) are short of issue slots, not of bandwidth.
__device__ __forceinline__ uint32_t bit_to_mask(uint32_t bits, uint32_t pos)
{
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "v"(pos));
    return m;
}
__device__ __forceinline__ uint32_t bit_to_mask_s(uint32_t uniform_bits, uint32_t pos)
{
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "s"(uniform_bits), "v"(pos));
    return m;
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t on, uint32_t off)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(on), "v"(off));
    return r;
}

// the four pixels of a lane: `word` = the plane word holding its rows, the rest wave-uniform (the chunk's descriptor).
// Tile codes: bit 0 = wall (white), bit 1 = goal and not wall (red)  SR:355-360, SR:288.
template <bool WIDE>      // WIDE: more than 16 tiles in a chunk (pu = 8), their codes take both words
__device__ __forceinline__ u32x4 top_chunk_pixels(const TopLane& L, uint32_t word, uint32_t s_lo, uint32_t s_hi,
                                                  int s_flags, uint32_t s_cm, int s_r0)
{
    const uint32_t ray_c = 0x00808080u, player_c = 0x00c0c0c0u, grid_c = 0x00ccccccu;   // SR:289-290, SR:364-367
    u32x4 o;
    if (s_flags & 2) {                                                       // wave-uniform: a tile's frame column
        o.x = o.y = o.z = o.w = grid_c;
    } else {
        uint32_t white, red;
        if (WIDE) {
            const uint32_t codes = L.code_hi ? s_hi : s_lo;
            white = bit_to_mask(codes, L.code_sh); red = bit_to_mask(codes, L.code_sh + 1);
        } else {
            white = bit_to_mask_s(s_lo, L.code_sh); red = bit_to_mask_s(s_lo, L.code_sh + 1);
        }
        const uint32_t fill = (white & 0x00FFFFFFu) | (red & 0x00FF0000u);
        o.x = L.first_row ? grid_c : fill;
        o.y = fill; o.z = fill;
        o.w = L.last_row ? grid_c : fill;
    }
    o.x = bfi(bit_to_mask(word, L.sh), ray_c, o.x);     o.y = bfi(bit_to_mask(word, L.sh + 1), ray_c, o.y);
    o.z = bfi(bit_to_mask(word, L.sh + 2), ray_c, o.z); o.w = bfi(bit_to_mask(word, L.sh + 3), ray_c, o.w);
    if (s_cm != 0u) {                                                        // a column of the player's circle
        const int q0 = L.r_lane - s_r0;                                      // mask bit of this lane's first pixel
        uint32_t cb = q0 >= 0 ? (q0 < 32 ? s_cm >> q0 : 0u) : (q0 > -4 ? s_cm << -q0 : 0u);
        o.x = bfi(bit_to_mask(cb, 0), player_c, o.x); o.y = bfi(bit_to_mask(cb, 1), player_c, o.y);
        o.z = bfi(bit_to_mask(cb, 2), player_c, o.z); o.w = bfi(bit_to_mask(cb, 3), player_c, o.w);
    }
    return o;
}

#ifdef RCW_DEV_SWITCHES   // (the experiment in front of the draw bodies: the store kernel's half)
constexpr int kFollowSpins = 400000;        // x (s_sleep 32 ~ 0.9 us + a load's round trip): about a second
// wave-uniform: returns when every block of agents up to agent `need`'s is complete; `have` = the leading complete blocks this wavefront knows of
__device__ __forceinline__ void top_follow_wait(const RcwDev& p, uint32_t need, uint32_t& have)
{
    const uint32_t sh = (uint32_t)p.top_blk_shift, need_blk = need >> sh;
    if (need_blk < have) return;
    const uint32_t nblk = ((uint32_t)p.B + (1u << sh) - 1u) >> sh, lane = threadIdx.x & 63u;
    for (int spins = 0; ; ++spins) {
        const uint32_t b = min(have + lane, nblk - 1u);                      // (lanes past the last block look at it again)
        const uint32_t agents = min(1u << sh, (uint32_t)p.B - (b << sh));
        uint32_t c = __hip_atomic_load(p.top_flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.top_debug & 32) c = __hip_atomic_fetch_add(p.top_flags + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (look with a read-modify-write)
        if (p.top_debug & 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const unsigned long long done = __ballot(c == p.top_epoch * agents);
        const uint32_t run = done == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~done);
        have = min(have + run, nblk);
        if (have > need_blk) {                                               // one acquire per advance, not per look: it invalidates the XCD's L2
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return;
        }
        if (run == 64u) continue;                                            // (all 64 complete: look at the next 64 at once)
        if (spins >= kFollowSpins) { p.err[0] = RCW_ERR_HIP; have = 0xFFFFFFFFu; return; }
        __builtin_amdgcn_s_sleep(32);
    }
}
#endif

// A group = the next 64 chunks of a wavefront; lane l holds the descriptor of the l-th.
struct TopGroup {
    int flags, r0, woff;         // bit 0 valid, bit 1 frame column | chunk row of the circle mask's bit 0 | plane word offset
    uint32_t code_lo, code_hi, cmask;
    uint32_t pw[8];              // register m of lane l = plane word (l & 7) of chunk 8 m + (l >> 3)
    int2 hd; uint32_t j, rb;     // (between issue and finish)
};
// first half: addresses and the loads (nothing here waits for a load)
__device__ __forceinline__ void top_group_issue(const RcwDev& p, const uint8_t* __restrict__ mask, uint32_t base, uint32_t G,
                                                uint32_t total, int lane, TopGroup& g)
{
    const int pu = p.pu, Wt = p.W * pu;
    const uint32_t k = (uint32_t)(p.H * pu) >> 8, wpu = (uint32_t)(p.H * pu) >> 5;
    const uint32_t id = base + (uint32_t)lane * G;
    bool valid = id < total;
    const uint32_t col = id / k, rb = id - col * k;
    const uint32_t a = col / (uint32_t)Wt, j = col - a * (uint32_t)Wt;
    const uint32_t tj = j / (uint32_t)pu, rj = j - tj * (uint32_t)pu;
    if (valid && mask != nullptr && mask[a] == 0) valid = false;
    g.flags = 0; g.woff = -1; g.code_lo = g.code_hi = 0u; g.hd = make_int2(0, 0); g.j = j; g.rb = rb;
    if (valid) {
        g.flags = 1 | ((rj == 0 || rj == (uint32_t)pu - 1) ? 2 : 0);         // SR:366-367: the tile's frame columns
        const uint2 cd = p.top_codes[((size_t)a * p.W + tj) * k + rb];
        g.hd = p.top_hdr[a];
        g.code_lo = cd.x; g.code_hi = cd.y;
        g.woff = (int)(col * wpu + rb * 8);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int wo = __shfl(g.woff, 8 * m + (lane >> 3), 64);
        g.pw[m] = wo >= 0 ? p.top_plane[(size_t)wo + (lane & 7)] : 0u;
    }
    if (p.top_parts > 1) {
        // several draw workgroups an agent OR their bits into this plane: it has to be zero when they start, and every word of it is read
        // exactly once, here — the reader leaves a zero behind (only words that hold a bit: most of a plane is zero already)
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int wo = __shfl(g.woff, 8 * m + (lane >> 3), 64);
            if (wo >= 0 && g.pw[m] != 0u) p.top_plane[(size_t)wo + (lane & 7)] = 0u;
        }
    }
}
// second half: everything that uses a loaded value.  The loads are waited for HERE, once per 64 chunks: left to the
// compiler the wait lands in every chunk's body as s_waitcnt vmcnt(0) (its wait-count bookkeeping merges the paths of
// the wave-uniform branches) — which also drains the wavefront's stores, one at a time.
__device__ __forceinline__ void top_group_finish(const RcwDev& p, const TopLane& L, TopGroup& g)
{
    const int rp = p.top_rp;
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("v_mov_b32 %0, %1" : "=v"(g.pw[m]) : "v"(g.pw[m]));
    asm volatile("v_mov_b32 %0, %1" : "=v"(g.code_lo) : "v"(g.code_lo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(g.code_hi) : "v"(g.code_hi));
    const int dist = (g.flags & 1) ? abs((int)g.j + 1 - g.hd.y) : 64;
    g.r0 = g.hd.x - 1 - rp - 256 * (int)g.rb;
    g.cmask = (uint32_t)__shfl((int)L.cm, dist & 63, 64);
    if (dist > rp || g.r0 >= 256 || g.r0 + 2 * rp < 0) g.cmask = 0u;
}

// Store kernel: the moving window of rcw_fill256_kernel over the top view's 1 KiB chunks (chunk id = flat pixel
// offset / 256: image column (a, j), row block rb).  Per 64 chunks of a wavefront, lane l computes the descriptor
// of the l-th (tile codes, frame column, the player's circle as a 32-bit row mask for that column), and the 8 plane
// words of each chunk are fetched by 8 lanes (8 loads per lane for the 64 chunks); per chunk the descriptor is
// broadcast with v_readlane, the plane words go through a wave-private 2 KiB of LDS, and lane l writes rows 4l..4l+3
// with one 16-byte store: colour = circle > ray line > tile frame > tile fill (SR:362-367, SR:473-477, SR:480).
// The circle: SD.Circle's pixels in the image column at distance c from the player's are the same rows relative to
// the player for every agent (midpoint circle, assumed): lane c computes that row mask once per kernel.
// (Issuing the next group's loads before this group's 64 stores, so that waiting for them would not wait for the
// stores, measured SLOWER: 201 vs 178 µs at 4096 x 256² px — the drain once per 64 chunks costs less than it looks.)
template <bool PLAIN, bool WIDE>
__global__ __launch_bounds__(kBlock) void rcw_top_store_kernel(const RcwDev p, const uint8_t* __restrict__ mask,
                                                               uint32_t chunk_begin, uint32_t chunk_end)   // the chunks of a run of agents
{
    const int lane = threadIdx.x & 63;
    const uint32_t G = gridDim.x * (kBlock / 64);
    const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the store's address is SGPR base + lane offset)
    const uint32_t total = chunk_end;
    u32x4* const out = reinterpret_cast<u32x4*>(p.top_view);
    const TopLane L = top_lane(p, lane);
    const size_t dstep = (size_t)G * 64;
    __shared__ uint32_t plane_words[(kBlock / 64) * 512];
    uint32_t* const lw_write = plane_words + (threadIdx.x >> 6) * 512 + lane;
    const uint32_t* const lw_read = plane_words + (threadIdx.x >> 6) * 512 + (lane >> 3);
    uint32_t base = chunk_begin + g;
    [[maybe_unused]] uint32_t have = 0u;                                     // (development experiment RCW_TOP_FOLLOW: blocks of agents known to be drawn)
    for (; base < total; base += G * 64) {
        TopGroup cur;
#ifdef RCW_DEV_SWITCHES
        if (p.top_follow) {                                                  // the agent of the group's last chunk
            const uint32_t id_last = min(base + 63u * G, total - 1u);
            top_follow_wait(p, id_last / (((uint32_t)(p.H * p.pu) >> 8) * (uint32_t)(p.W * p.pu)), have);
        }
#endif
        top_group_issue(p, mask, base, G, total, lane, cur);
        top_group_finish(p, L, cur);
        // the plane words go through a wave-private 2 KiB of LDS (index 8 t + word: lane l's register m is entry
        // 64 m + l), so that ONE loop over the 64 chunks can fetch them (a register per eight chunks would need eight
        // copies of the loop — and the compiler then carries all their store pointers through every one of them)
        __builtin_amdgcn_wave_barrier();                                     // (the lanes of a wavefront exchange through it: no reordering across)
#pragma unroll
        for (int m = 0; m < 8; ++m) lw_write[64 * m] = cur.pw[m];
        __builtin_amdgcn_wave_barrier();
        u32x4* dst = out + (size_t)base * 64;                                // wave-uniform
        asm volatile(".p2align 6");          // the chunk loop starts on an instruction-cache line: its speed moved by 2 % (159.5 / 163 µs) with the code before it
#pragma unroll 2                                                             // (1: 173 us, 2 / 4 / 8: 169; reading the next chunk's word one chunk ahead: 170)
        for (int t = 0; t < 64; ++t, dst += dstep) {
            {
                const int s_flags = __builtin_amdgcn_readlane(cur.flags, t);
                if (!(s_flags & 1)) continue;                                // wave-uniform: past the end / masked out
                const uint32_t w = lw_read[8 * t];
                const u32x4 o = top_chunk_pixels<WIDE>(L, w, (uint32_t)__builtin_amdgcn_readlane((int)cur.code_lo, t),
                                                       WIDE ? (uint32_t)__builtin_amdgcn_readlane((int)cur.code_hi, t) : 0u, s_flags,
                                                       (uint32_t)__builtin_amdgcn_readlane((int)cur.cmask, t),
                                                       __builtin_amdgcn_readlane(cur.r0, t));
                store16<PLAIN>(dst + lane, o);
            }
        }
    }
}

// The same sweep for image heights that are a multiple of 128, 64 or 32 rows but not of 256 (and tiles that divide that
// number): a 1 KiB chunk then holds U = 2, 4 or 8 UNITS — runs of 128 / 64 / 32 rows of one image column — which may
// belong to different columns, so the descriptor is per unit: lane l of the prefetch computes the U descriptors of its
// chunk and parks them, like the plane words, in wave-private LDS; in the chunk loop a lane reads its unit's
// (lane / (64 / U)) back with one ds_read (broadcast with v_readlane and picked with selects instead: the same at
// U = 2 and 4, 244 instead of 212 µs at U = 8).  The plane needs nothing new: unpadded, its bit index IS the flat pixel
// index, so a chunk's plane words are 8 consecutive ones whatever the columns.
// A unit's descriptor word: bits 0..27 the 2-bit fill codes of its (at most 14) tiles, bit 28 frame column, bit 31 valid.
// (Issuing all U units' loads before the first use, one wait instead of U, changes nothing measurable.)
template <bool PLAIN, int U>
__global__ __launch_bounds__(kBlock) void rcw_top_store_units_kernel(const RcwDev p, const uint8_t* __restrict__ mask,
                                                                     uint32_t chunk_begin, uint32_t chunk_end)
{
    constexpr int LPU = 64 / U, UPX = 256 / U;                               // lanes, pixels of a unit
    constexpr int kWaveWords = 512 + 3 * 64 * U;                             // plane words | descriptors | circle masks | circle rows
    const int lane = threadIdx.x & 63;
    const uint32_t G = gridDim.x * (kBlock / 64);
    const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pu = p.pu, Ht = p.H * pu, Wt = p.W * pu, rp = p.top_rp;
    const uint32_t k = (uint32_t)Ht / UPX;                                   // units of an image column
    const uint32_t total_units = (uint32_t)p.B * (uint32_t)Wt * k;
    const uint32_t total = chunk_end;                                        // (the batch's last chunk may be short: total_units)
    const uint32_t ray_c = 0x00808080u, player_c = 0x00c0c0c0u, grid_c = 0x00ccccccu;   // SR:289-290, SR:364-367
    u32x4* const out = reinterpret_cast<u32x4*>(p.top_view);
    const TopLane L = top_lane(p, lane);                                     // (only its circle table and plane-bit shift apply here)
    const int u_lane = lane / LPU, r_lane = (lane - u_lane * LPU) * 4;       // this lane's unit of the chunk, its first row in it
    const int tl = r_lane / pu, ri = r_lane - tl * pu;
    const bool first_row = ri == 0, last_row = ri + 3 == pu - 1;             // SR:364-365
    const uint32_t code_sh = 2u * (uint32_t)tl;
    const size_t dstep = (size_t)G * 64;
    __shared__ uint32_t wave_words[(kBlock / 64) * kWaveWords];
    uint32_t* const ws = wave_words + (threadIdx.x >> 6) * kWaveWords;
    uint32_t* const lw_write = ws + lane;
    const uint32_t* const lw_read = ws + (lane >> 3);
    uint32_t* const desc = ws + 512;                                         // [64 chunks][U]
    uint32_t* const circ = desc + 64 * U;
    uint32_t* const crow = circ + 64 * U;
    [[maybe_unused]] uint32_t have = 0u;                                     // (development experiment RCW_TOP_FOLLOW: blocks of agents known to be drawn)
    for (uint32_t base = chunk_begin + g; base < total; base += G * 64) {
#ifdef RCW_DEV_SWITCHES
        if (p.top_follow) {                                                  // the agent of the last unit of the group's last chunk
            const uint32_t id_last = min(base + 63u * G, total - 1u);
            const uint32_t un_last = min(id_last * U + (U - 1), total_units - 1u);
            top_follow_wait(p, un_last / (k * (uint32_t)Wt), have);
        }
#endif
        const uint32_t id = base + (uint32_t)lane * G;
        uint32_t packed[U], cmask[U];
        int r0[U];
        int nvalid = 0;
        bool any_circle = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t un = id * U + u;
            bool valid = id < total && un < total_units;
            const uint32_t col = un / k, rb = un - col * k;
            const uint32_t a = col / (uint32_t)Wt, j = col - a * (uint32_t)Wt;
            const uint32_t tj = j / (uint32_t)pu, rj = j - tj * (uint32_t)pu;
            if (valid && mask != nullptr && mask[a] == 0) valid = false;
            packed[u] = 0u; r0[u] = 0;
            int dist = 64;
            if (valid) {
                const uint2 cd = p.top_codes[((size_t)a * p.W + tj) * k + rb];
                const int2 hd = p.top_hdr[a];
                packed[u] = (cd.x & 0x0FFFFFFFu) | ((rj == 0 || rj == (uint32_t)pu - 1) ? 1u << 28 : 0u) | (1u << 31);
                dist = abs((int)j + 1 - hd.y);
                r0[u] = hd.x - 1 - rp - UPX * (int)rb;                       // unit row of the circle mask's bit 0
                nvalid += 1;
            }
            uint32_t c = (uint32_t)__shfl((int)L.cm, dist & 63, 64);
            if (dist > rp || r0[u] >= UPX || r0[u] + 2 * rp < 0) c = 0u;
            cmask[u] = c;
            any_circle = any_circle || c != 0u;
        }
        const int state_l = (nvalid > 0 ? 1 : 0) | (nvalid == U ? 2 : 0) | (any_circle ? 4 : 0);
        const int woff_l = nvalid > 0 ? (int)(id * 8u) : -1;                 // plane word = flat pixel / 32
        uint32_t pw[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int wo = __shfl(woff_l, 8 * m + (lane >> 3), 64);
            pw[m] = wo >= 0 ? p.top_plane[(size_t)wo + (lane & 7)] : 0u;
        }
        // (the loads are waited for here, by the LDS writes that use them — once per 64 chunks, see top_group_finish)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 8; ++m) lw_write[64 * m] = pw[m];
#pragma unroll
        for (int u = 0; u < U; ++u) { desc[lane * U + u] = packed[u]; circ[lane * U + u] = cmask[u]; crow[lane * U + u] = (uint32_t)r0[u]; }
        __builtin_amdgcn_wave_barrier();
        u32x4* dst = out + (size_t)base * 64;
        asm volatile(".p2align 6");          // (as in rcw_top_store_kernel)
#pragma unroll 2
        for (int t = 0; t < 64; ++t, dst += dstep) {
            const int s_state = __builtin_amdgcn_readlane(state_l, t);
            if (!(s_state & 1)) continue;                                    // wave-uniform: past the end / masked out
            const uint32_t w = lw_read[8 * t];
            const uint32_t pk = desc[t * U + u_lane];
            const uint32_t fill = (bit_to_mask(pk, code_sh) & 0x00FFFFFFu) | (bit_to_mask(pk, code_sh + 1) & 0x00FF0000u);
            const uint32_t frame = bit_to_mask(pk, 28);                      // SR:366-367: the tile's frame columns
            u32x4 o;
            o.x = bfi(frame, grid_c, first_row ? grid_c : fill);
            o.y = bfi(frame, grid_c, fill);
            o.z = o.y;
            o.w = bfi(frame, grid_c, last_row ? grid_c : fill);
            o.x = bfi(bit_to_mask(w, L.sh), ray_c, o.x);     o.y = bfi(bit_to_mask(w, L.sh + 1), ray_c, o.y);
            o.z = bfi(bit_to_mask(w, L.sh + 2), ray_c, o.z); o.w = bfi(bit_to_mask(w, L.sh + 3), ray_c, o.w);
            if (s_state & 4) {                                               // some unit of the chunk crosses the player's circle
                const uint32_t cmv = circ[t * U + u_lane];
                const int q0 = r_lane - (int)crow[t * U + u_lane];
                const uint32_t cb = q0 >= 0 ? (q0 < 32 ? cmv >> q0 : 0u) : (q0 > -4 ? cmv << -q0 : 0u);
                o.x = bfi(bit_to_mask(cb, 0), player_c, o.x); o.y = bfi(bit_to_mask(cb, 1), player_c, o.y);
                o.z = bfi(bit_to_mask(cb, 2), player_c, o.z); o.w = bfi(bit_to_mask(cb, 3), player_c, o.w);
            }
            if (s_state & 2) store16<PLAIN>(dst + lane, o);                  // every unit of the chunk is written
            else if (pk >> 31) store16<PLAIN>(dst + lane, o);                // a chunk at the end / at a masked agent's border
        }
    }
}


// The moving-window store for ANY pixel scale from 9 pixels a tile and any image of at least 42 rows whose height is a
// multiple of 4 (pu_per_tu and the map size are free kwargs, SR:260-261, SR:269): a chunk is 256 consecutive pixels of
// the flat (H·pu, W·pu, B) batch, whatever image columns — of one agent or two — they belong to, so every wavefront
// store is an aligned 1 KiB.  As in rcw_fill_flat_kernel, lane l of the prefetch finds (first column, row in it) of
// the wavefront's l-th next chunk and parks one 16-byte descriptor per touched column in wave-private LDS:
//   x: bit 31 valid | 30 frame column (SR:366-367) | 29 a column of the player's circle | 28..16 its distance from the
//      player's column | 15..0 the tile row of the code window's first tile
//   y, z: the 2-bit tile_map entries (bit 0 WALL, bit 1 GOAL: BitArray{3}(2, H, W) read as it lies in HBM, SR:54) of
//      the 32 tiles of this image column from that tile row on — more than a 256-row run can touch from 9 pixels a tile
//   w: the image row of the circle mask's bit 0 (ip - 1 - rp)
// In the chunk loop a lane finds its column without a division (flat_locate), and what depends on its row alone — the
// tile row, which of its four pixels lie on a tile's frame rows (SR:364-365) and which in the following tile (pixel
// scales that are not a multiple of 4) — in a table the workgroup builds once in LDS (one word per four rows).  It
// reads its column's descriptor with one ds_read_b128 and its plane word, resolves circle > ray line > tile frame >
// tile fill (SR:362-367, SR:473-477, SR:480) with bit-field extracts and inserts, no compare, and writes its four
// pixels with one 16-byte store.  A group whose 64 chunks are all whole and unmasked fetches the next chunk's three LDS
// values while it computes this one's pixels.  The circle of any radius: SD.Circle's rows at column distance c are the
// same for every agent (midpoint circle, assumed); the workgroup tabulates them in LDS once, as bit rows.
template <int POS>
__device__ __forceinline__ uint32_t bit_to_mask_c(uint32_t bits)                     // v_bfe_i32 with an inline-constant position
{
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(POS));
    return m;
}
// The row table (one entry per four image rows, i.e. per lane group): the tile row of the group's first pixel, and as
// BYTE masks (0xFF / 0x00 in byte e for pixel e) which of the four pixels lie in the following tile and which on a
// tile's first or last row (SR:364-365).
__device__ __forceinline__ void top_row_entry(int r, int pu, uint32_t& ti_out, uint32_t& next_bytes, uint32_t& grid_bytes)
{
    const int ti = r / pu, ri = r - ti * pu;
    ti_out = (uint32_t)ti; next_bytes = 0u; grid_bytes = 0u;
    for (int e = 0; e < 4; ++e) {
        const bool next = ri + e >= pu;
        const int rie = next ? ri + e - pu : ri + e;
        if (next) next_bytes |= 0xFFu << (8 * e);
        if (rie == 0 || rie == pu - 1) grid_bytes |= 0xFFu << (8 * e);
    }
}
// (four bits -> four byte masks: n · 0x204081 puts bit e at bit 8 e — the four shifted copies do not overlap — and
// v_perm_b32's selector 0x0C yields the byte 0x00, 0x0D the byte 0xFF: nibble_to_bytes below)
// All the top view's colours but one are greys (SR:288-290, SR:364-367: black, white, 0xcccccc grid, 0x808080 ray,
// 0xc0c0c0 player) and the goal tile's red is 0xFF0000: a lane carries its four pixels as two packed words, R (the red
// byte of each pixel) and GB (the byte that is both green and blue), so that every overlay is two v_bfi_b32 for all
// four pixels — colour = circle > ray line > tile frame > tile fill (SR:362-367, SR:473-477, SR:480) — and one
// v_perm_b32 per pixel unpacks them at the end.
struct TopFlatConst { uint32_t sh0; int rp, cwt; const uint32_t* ctab; uint32_t k01, k0c; };
// four bits -> four byte masks with the two constants in registers (v_mul_u32_u24, v_and_or_b32, v_perm_b32)
__device__ __forceinline__ uint32_t nibble_to_bytes(const TopFlatConst& C, uint32_t n)
{
    return __builtin_amdgcn_perm(0u, 0u, (__umul24(n, 0x00204081u) & C.k01) | C.k0c);
}
// NARROW: a 256-row run touches at most 16 tiles (pu >= 19): the code window is one word (d.y) and d.z is the frame-column
// word (0 / ~0); else the window is d.y | d.z << 32 and the frame column is bit 30 of d.x.
template <bool STRADDLE, bool NARROW>
__device__ __forceinline__ u32x4 top_flat_pixels(const TopFlatConst& C, int r, uint4 d, uint32_t w, uint32_t ti, uint32_t next_bytes,
                                                 uint32_t grid_bytes, bool circle_chunk)
{
    const int trel = (int)ti - (int)(d.x & 0xFFFFu);
    // this tile's 2 bits, then the next one's
    const uint32_t c4 = NARROW ? d.y >> (2 * trel) : (uint32_t)((((unsigned long long)d.z << 32) | d.y) >> (2 * trel));
    // tile fill: WALL (bit 0) white before GOAL (bit 1) red, else black  SR:355-360, colours SR:288
    uint32_t GB = bit_to_mask_c<0>(c4);                                      // all four bytes alike: 0xFF where white
    uint32_t R = GB | bit_to_mask_c<1>(c4);
    if (STRADDLE) {
        const uint32_t GB1 = bit_to_mask_c<2>(c4), R1 = GB1 | bit_to_mask_c<3>(c4);
        GB = bfi(next_bytes, GB1, GB); R = bfi(next_bytes, R1, R);
    }
    const uint32_t gm = grid_bytes | (NARROW ? d.z : bit_to_mask_c<30>(d.x));   // frame rows SR:364-365 | the tile's frame columns SR:366-367
    GB = bfi(gm, 0xCCCCCCCCu, GB); R = bfi(gm, 0xCCCCCCCCu, R);
    const uint32_t rm = nibble_to_bytes(C, __builtin_amdgcn_ubfe(w, C.sh0, 4));   // ray lines SR:473-477
    GB = bfi(rm, 0x80808080u, GB); R = bfi(rm, 0x80808080u, R);
    if (circle_chunk) {                                                      // wave-uniform: some column of the chunk crosses the player's circle
        const int q0 = r - (int)d.w;                                         // mask bit of this lane's first pixel
        if ((d.x & 0x20000000u) && q0 > -4 && q0 <= 2 * C.rp) {
            const int bidx = q0 + 32;                                        // (the row's leading zero word absorbs q0 < 0)
            const uint32_t* const row = C.ctab + ((d.x >> 16) & 0x1FFFu) * C.cwt + (bidx >> 5);
            const uint32_t cb = (uint32_t)((((unsigned long long)row[1] << 32) | row[0]) >> (bidx & 31));
            const uint32_t cm = nibble_to_bytes(C, cb & 15u);
            GB = bfi(cm, 0xC0C0C0C0u, GB); R = bfi(cm, 0xC0C0C0C0u, R);
        }
    }
    u32x4 o;                                                                 // pixel e = 0x00 | R[e] | GB[e] | GB[e]
    o.x = __builtin_amdgcn_perm(R, GB, 0x0C040000u); o.y = __builtin_amdgcn_perm(R, GB, 0x0C050101u);
    o.z = __builtin_amdgcn_perm(R, GB, 0x0C060202u); o.w = __builtin_amdgcn_perm(R, GB, 0x0C070303u);
    return o;
}

// Global loads whose completion the COMPILER does not track (the store kernels' descriptor prefetch).  gfx9 counts loads
// and stores in one in-order counter (vmcnt): the wait the compiler inserts before the first use of a loaded value that
// was issued ahead of a loop of stores is vmcnt(0..few) — it waits for the loads AND drains every store behind them,
// and the descriptor arithmetic that follows then runs with nothing of this wavefront in flight; all wavefronts do so
// at the same moments (they run in lockstep, which the moving window needs), so the memory system idles through it.
// Issued like this and awaited with flat_wait_loads<63>() — "at most 63 operations outstanding" = everything older
// than the 63 newest, i.e. all loads that were followed by at least 63 stores — the stores stay in flight while the
// descriptors are made.  The destination registers must not be read or copied between the load and the wait: they are
// written and awaited inside ONE loop iteration (no loop-carried copies), and the wait names them as in/out operands.
[[maybe_unused]] __device__ __forceinline__ void flat_load_b32(uint32_t& dst, const uint32_t* addr) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(addr) : "memory"); }
__device__ __forceinline__ void flat_load_u8(uint32_t& dst, const uint8_t* addr) { asm volatile("global_load_ubyte %0, %1, off" : "=v"(dst) : "v"(addr) : "memory"); }
__device__ __forceinline__ void flat_load_b64(unsigned long long& dst, const void* addr) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory"); }
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void flat_load_b96(u32x3& dst, const void* addr) { asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory"); }
__device__ __forceinline__ void flat_load_b128(u32x4& dst, const void* addr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory"); }
// floor(n / d) as fast_div, for n, d < 2^15: the products fit v_mul_u32_u24 (full rate; v_mul_lo_u32 takes four passes)
__device__ __forceinline__ int fast_div24(int n, int d, float inv_d)
{
    int q = (int)((float)n * inv_d);
    q -= ((int)__umul24((uint32_t)q, (uint32_t)d) > n) ? 1 : 0;
    q += ((int)__umul24((uint32_t)(q + 1), (uint32_t)d) <= n) ? 1 : 0;
    return q;
}

// what a lane holds of the wavefront's l-th next chunk between the loads and their use (one group ahead); K = the image
// columns a chunk may touch (a template parameter: every load below is unconditional straight-line code, see flat_load_b32)
template <int K>
struct TopFlatPre {
    uint32_t rem, j0, a0;                // the chunk's first pixel: row in its image column, that column, its agent
    int touched, ti_first;               // its last column (relative; -1: no such chunk), the tile row of its first pixel
    bool full, two;                      // all 256 pixels lie inside the batch; they belong to two agents
    unsigned long long hd[2]; uint32_t mk[2];   // player pixel (ip | jp << 32) / mask byte of the first pixel's agent and of the following one
    u32x3 tw[K];                         // three tile_map words from each touched column's code window on
    u32x4 pa[2], pb[2];                  // the chunk's 8 plane words in the first pixel's agent's region and in the following agent's
};
// wait until at most N vector-memory operations are outstanding; names every loaded register as in/out
template <int N, int K>
__device__ __forceinline__ void flat_wait_loads(TopFlatPre<K>& P)
{
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(P.hd[0]), "+v"(P.hd[1]), "+v"(P.mk[0]), "+v"(P.mk[1]), "+v"(P.pa[0]), "+v"(P.pa[1]), "+v"(P.pb[0]), "+v"(P.pb[1])
                 : "n"(N) : "memory");
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(P.tw[j]) :: "memory");
}

template <bool STRADDLE, bool NARROW, int K>
__global__ __launch_bounds__(kBlock) void rcw_top_store_flat_kernel(const RcwDev p, const uint8_t* __restrict__ mask,
                                                                    uint32_t chunk_begin, uint32_t chunk_end,
                                                                    int agent_lo, int agent_hi)
{
    constexpr bool PLAIN = false;                                          // (non-temporal stores, as every window kernel)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t G = gridDim.x * (kBlock / 64);
    const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    const int pu = p.pu, Ht = p.H * pu, Wt = p.W * pu, rp = p.top_rp;
    constexpr int KS = K;
    const float inv_pu = 1.0f / (float)pu;
    const unsigned px_agent = (unsigned)Ht * (unsigned)Wt;
    const unsigned long long total_px = (unsigned long long)p.B * px_agent;
    const unsigned PW = (unsigned)p.top_plane_words;
    // LDS: [4 wavefronts][512 plane words] | [4 wavefronts][64 chunks][KS] descriptors | the circle's bit rows | the row table
    uint32_t* const lw = lds + wave * 512;
    const uint32_t* const lw_read = lw + (lane >> 3);
    uint4* const desc = reinterpret_cast<uint4*>(lds + (kBlock / 64) * 512) + (size_t)wave * 64 * KS;
    uint32_t* const ctab = lds + (kBlock / 64) * 512 + (size_t)(kBlock / 64) * 64 * KS * 4;
    const int cnw = (2 * rp + 1 + 31) / 32, cwt = cnw + 2;                 // a row: [zero word | 2 rp + 1 mask bits | zero word]
    // the row table: one 16-byte entry per four rows (tile row | next-tile bytes | frame-row bytes | -), behind the circle rows
    uint4* const rtab = reinterpret_cast<uint4*>(ctab + (((rp + 1) * cwt + 3) & ~3));
    for (int k = threadIdx.x; k < (rp + 1) * cwt; k += kBlock) ctab[k] = 0u;
    for (int k = threadIdx.x; k < (Ht >> 2); k += kBlock) { uint4 e; e.w = 0u; top_row_entry(4 * k, pu, e.x, e.y, e.z); rtab[k] = e; }
    __syncthreads();
    for (int c = threadIdx.x; c <= rp; c += kBlock) {                      // SD.Circle SR:480 (midpoint circle, assumed)
        uint32_t* const row = ctab + c * cwt + 1;
        auto set = [&](int off) { const int q = rp + off; row[q >> 5] |= 1u << (q & 31); };
        int x = 0, y = rp, dd = 1 - rp;
        while (x <= y) {
            if (y == c) { set(x); set(-x); }
            if (x == c) { set(y); set(-y); }
            x += 1;
            if (dd < 0) dd += 2 * x + 1;
            else { y -= 1; dd += 2 * (x - y) + 1; }
        }
    }
    __syncthreads();
    const FlatLane L = flat_lane(lane, Ht);
    TopFlatConst C;
    C.sh0 = (uint32_t)(4 * lane) & 31u; C.rp = rp; C.cwt = cwt; C.ctab = ctab; C.k01 = 0x01010101u; C.k0c = 0x0C0C0C0Cu;
    asm volatile("" : "+v"(C.k01), "+v"(C.k0c));                             // (in registers: two literals do not fit one v_and_or_b32)
    const uint32_t lane16 = (uint32_t)lane * 16u;                            // the store's address: uniform chunk base + this
    // this lane's chunk of the first group as (image column of the flat batch, row in it); every group moves all lanes alike
    const unsigned long long id0 = (unsigned long long)chunk_begin + g + (unsigned long long)lane * G;
    // The wavefront -> chunk assignment TURNS by R slots from group to group: wavefront g takes slot (g + k R) mod G of group k — every
    // group is still one compact window, every chunk is written once.  Without it (R = 0), where an image is a whole number of chunks
    // that divides G (128 x 128 px = 64 chunks, 128 x 256 = 128), a wavefront meets the SAME image columns of every agent for the whole
    // launch, and the kernel takes 10-14 % longer (round 4: 190 -> 172 us / GiB at 8x8 tiles of 16 px, 203 -> 179 at 4x4 of 32; a turn
    // of 64 slots — the same columns again — changes nothing; images that are no such number of chunks are not affected: DESIGN.md §4.4).
    const uint32_t R = (uint32_t)p.top_rotate % G;                           // (33 slots; the development build reads RCW_TOP_ROTATE)
    const unsigned long long step_px = ((unsigned long long)G * 64 + R) * 256;        // a lane's step from group to group ...
    const unsigned long long step_px_w = step_px - (unsigned long long)G * 256;       // ... and where its slot wraps past G
    const uint32_t dq = (uint32_t)(step_px / (unsigned)Ht), dr = (uint32_t)(step_px - (unsigned long long)dq * (unsigned)Ht);
    const uint32_t dq_w = (uint32_t)(step_px_w / (unsigned)Ht), dr_w = (uint32_t)(step_px_w - (unsigned long long)dq_w * (unsigned)Ht);
    uint32_t slot_i = g;                                                               // the slot of the group `issue` is asked for next
    uint32_t col = (uint32_t)((id0 * 256) / (unsigned)Ht);
    uint32_t rem = (uint32_t)(id0 * 256 - (unsigned long long)col * (unsigned)Ht);
    u32x4* const out4 = reinterpret_cast<u32x4*>(p.top_view);
    const size_t dstep = (size_t)G * 64;
    const uint32_t last_agent = (uint32_t)p.B - 1u, last_word = (uint32_t)p.nwords - 1u;

    // The loads of a group: every address is clamped into its array instead of the load being predicated (a predicated
    // load is a branch around it), nothing here waits.  They are issued ONE GROUP AHEAD — before the 64 stores of the
    // current group — so that their latency passes while the wavefront stores, and awaited with flat_wait_loads.  What a
    // wavefront does between two groups' stores is time the whole chip spends not storing (the wavefronts run in lockstep):
    // (column, agent) of a lane's next chunk are carried from group to group, tile rows come from the row table, the three
    // tile_map words of a column are one 12-byte load, a chunk's plane words two 16-byte loads by the chunk's own lane.
    uint32_t a_cur = col / (unsigned)Wt, j_cur = col - a_cur * (unsigned)Wt;      // (agent, image column) of this lane's next chunk
    const uint32_t dqa = dq / (unsigned)Wt, dqj = dq - dqa * (unsigned)Wt;        // ... move by this much a group (+ 1 column on a row wrap)
    const uint32_t dqa_w = dq_w / (unsigned)Wt, dqj_w = dq_w - dqa_w * (unsigned)Wt;
    [[maybe_unused]] uint32_t have = 0u;                                     // (development experiment RCW_TOP_FOLLOW: blocks of agents known to be drawn)
    auto issue = [&](uint32_t base, TopFlatPre<K>& P) {
        const uint32_t id = base + (uint32_t)lane * G;
        const bool exists = id < chunk_end;
#ifdef RCW_DEV_SWITCHES
        if (p.top_follow)                                                    // the last agent any of the group's chunks touches (the assignment turns: no lane order)
            top_follow_wait(p, (uint32_t)__builtin_amdgcn_readlane(wave_max_in_lane63(exists ? (int)min(a_cur + 1u, last_agent) : 0), 63), have);
#endif
        P.rem = rem; P.a0 = a_cur; P.j0 = j_cur;
        int touched = 0;
#pragma unroll
        for (int k = 1; k < K; ++k) touched += (rem + 255u >= (unsigned)(k * Ht)) ? 1 : 0;
        P.touched = exists ? touched : -1;
        P.ti_first = (int)rtab[rem >> 2].x;                                  // (rem is a multiple of 4)
        P.full = ((unsigned long long)id + 1) * 256 <= total_px;
        P.two = P.j0 + (unsigned)touched >= (unsigned)Wt;
        const uint32_t a0c = min(P.a0, last_agent), a1c = min(P.a0 + 1u, last_agent);
        flat_load_b64(P.hd[0], p.top_hdr + a0c); flat_load_b64(P.hd[1], p.top_hdr + a1c);
        P.mk[0] = P.mk[1] = 1u;
        if (mask != nullptr) { flat_load_u8(P.mk[0], mask + a0c); flat_load_u8(P.mk[1], mask + a1c); }   // wave-uniform
        const uint32_t* const tm0 = p.tile_map + (size_t)a0c * p.nwords;
        const uint32_t* const tm1 = p.tile_map + (size_t)a1c * p.nwords;
        int tj = fast_div24((int)P.j0, pu, inv_pu), rj = (int)P.j0 - (int)__umul24((uint32_t)tj, (uint32_t)pu);
        uint32_t jx = P.j0;
        const uint32_t* tm = tm0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t wi = (__umul24((uint32_t)p.H, (uint32_t)tj) + (j == 0 ? (uint32_t)P.ti_first : 0u)) >> 4;   // word of the code window's first tile
            flat_load_b96(P.tw[j], tm + min(wi, last_word));                 // (two words of slack lie behind the last agent's map)
            jx += 1; rj += 1;
            if (rj == pu) { rj = 0; tj += 1; }
            if (jx == (unsigned)Wt) { jx = 0; tj = 0; rj = 0; tm = tm1; }   // the following agent's first column
        }
        // the chunk's 8 plane words in the region of the first pixel's agent, and (read as zero unless the chunk straddles
        // two agents) the first 8 of the following agent's region
        const uint32_t c0 = (uint32_t)(((unsigned long long)a0c * px_agent) >> 8);
        const uint32_t* const pwa = p.top_plane + (size_t)a0c * PW + (size_t)(exists ? id - c0 : 0u) * 8u;
        const uint32_t* const pwb = p.top_plane + (size_t)a1c * PW;
        flat_load_b128(P.pa[0], pwa); flat_load_b128(P.pa[1], pwa + 4);
        flat_load_b128(P.pb[0], pwb); flat_load_b128(P.pb[1], pwb + 4);
        // this lane's chunk of the next group
        const bool wrap = slot_i + R >= G;                                   // (wave-uniform)
        slot_i = wrap ? slot_i + R - G : slot_i + R;
        col += wrap ? dq_w : dq; rem += wrap ? dr_w : dr;
        uint32_t jn = j_cur + (wrap ? dqj_w : dqj);
        if (rem >= (unsigned)Ht) { rem -= (unsigned)Ht; col += 1; jn += 1; }
        a_cur += wrap ? dqa_w : dqa;
        if (jn >= (unsigned)Wt) { jn -= (unsigned)Wt; a_cur += 1; }
        j_cur = jn;
    };
    // ... and their use: the descriptors and plane words of the group into wave-private LDS
    auto finish = [&](const TopFlatPre<K>& P, int& state_l, int& rem_l) {
        bool all_valid = P.touched >= 0 && P.full, any_circle = false;
        int tj = fast_div24((int)P.j0, pu, inv_pu), rj = (int)P.j0 - (int)__umul24((uint32_t)tj, (uint32_t)pu);
        uint32_t jx = P.j0, a = P.a0;
        int second = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool valid = j <= P.touched && (int)a >= agent_lo && (int)a < agent_hi && (second ? P.mk[1] : P.mk[0]) != 0u;
            if (j <= P.touched && !valid) all_valid = false;
            const int ti_lo = j == 0 ? P.ti_first : 0;
            const int sh = (int)((__umul24((uint32_t)p.H, (uint32_t)tj) + (uint32_t)ti_lo) & 15u) * 2;
            // ({w1, w0} >> sh) and ({w2, w1} >> sh), sh < 32: v_alignbit_b32 by name (written as 64-bit shifts the compiler
            // pairs the words up in registers where they are LOADED, with a copy of w1 in between)
            uint32_t lo, hi = 0u;
            asm("v_alignbit_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(P.tw[j].y), "v"(P.tw[j].x), "v"(sh));
            if (!NARROW) asm("v_alignbit_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(P.tw[j].z), "v"(P.tw[j].y), "v"(sh));
            const unsigned long long hd = second ? P.hd[1] : P.hd[0];
            const int dist = abs((int)jx + 1 - (int)(hd >> 32)), r0 = (int)(uint32_t)hd - 1 - rp;
            const bool circle = valid && dist <= rp;
            any_circle = any_circle || circle;
            const bool frame = rj == 0 || rj == pu - 1;
            const uint32_t x = (valid ? 0x80000000u : 0u) | (frame ? 0x40000000u : 0u) |
                               (circle ? 0x20000000u : 0u) | ((uint32_t)(circle ? dist : 0) << 16) | (uint32_t)ti_lo;
            desc[lane * KS + j] = make_uint4(x, lo, NARROW ? (frame ? 0xFFFFFFFFu : 0u) : hi, (uint32_t)r0);
            jx += 1; rj += 1;
            if (rj == pu) { rj = 0; tj += 1; }
            if (jx == (unsigned)Wt) { jx = 0; tj = 0; rj = 0; a += 1; second = 1; }
        }
        const uint32_t mb = (P.two && P.a0 + 1u <= last_agent) ? 0xFFFFFFFFu : 0u;
        u32x4* const lw4 = reinterpret_cast<u32x4*>(lw + lane * 8);          // plane word w of the wavefront's chunk c: lw[8 c + w]
        lw4[0] = P.pa[0] | (P.pb[0] & mb); lw4[1] = P.pa[1] | (P.pb[1] & mb);
        state_l = (P.touched >= 0 ? 1 : 0) | (all_valid ? 2 : 0) | (any_circle ? 4 : 0);
        rem_l = (int)P.rem;
    };

    // the 64 chunks of a group: descriptors, plane words and the row table from LDS -> pixels -> stores.  Returns whether
    // exactly 64 stores went out (the branch-free loop), which is what flat_wait_loads<63> may rely on.
    auto store_group = [&](uint32_t base, int state_l, int rem_l) -> bool {
        char* dst = reinterpret_cast<char*>(out4) + (size_t)base * 1024;     // wave-uniform: the chunk's first byte
        const size_t dstep_b = dstep * 16;
        auto put = [&](const u32x4& o) { store16<PLAIN>(reinterpret_cast<u32x4*>(dst + lane16), o); };
        const unsigned long long whole = __ballot((state_l & 3) == 3);
        const unsigned long long circle_chunks = __ballot((state_l & 4) != 0);   // bit t: chunk t crosses a player's circle
        if (whole == ~0ull) {
            // every chunk of the group is whole and unmasked: the next chunk's LDS values on their way while this one's pixels are made
            int rel, r, rel_n, r_n;
            flat_locate(L, __builtin_amdgcn_readlane(rem_l, 0), Ht, rel, r);
            uint4 d = desc[rel];
            uint32_t w = lw_read[0];
            uint4 re = rtab[r >> 2];
#pragma unroll 2
            for (int t = 0; t < 64; ++t, dst += dstep_b) {
                // (the last trip fetches a 65th chunk's values: lane 0's row again — v_readlane takes the lane number modulo
                // 64 — and whatever lies behind this wavefront's descriptors and plane words in the workgroup's LDS; unused)
                flat_locate(L, __builtin_amdgcn_readlane(rem_l, t + 1), Ht, rel_n, r_n);
                const uint4 d_n = desc[(t + 1) * KS + rel_n];
                const uint4 re_n = rtab[r_n >> 2];
                const uint32_t w_n = lw_read[8 * (t + 1)];
                put(top_flat_pixels<STRADDLE, NARROW>(C, r, d, w, re.x, re.y, re.z, ((circle_chunks >> t) & 1ull) != 0));
                d = d_n; w = w_n; re = re_n; r = r_n;
            }
            return true;
        }
        // a masked agent's or a run's border, or the batch's end, lies in the group: its LEADING whole chunks (all up to the
        // batch's last chunk, in the last group of every wavefront) take the same loop, the rest the general one
        const int n_fast = (int)__builtin_ctzll(~whole);
        int t0 = 0;
        if (n_fast >= 4) {
            int rel, r, rel_n, r_n;
            flat_locate(L, __builtin_amdgcn_readlane(rem_l, 0), Ht, rel, r);
            uint4 d = desc[rel];
            uint32_t w = lw_read[0];
            uint4 re = rtab[r >> 2];
#pragma unroll 1
            for (int t = 0; t < n_fast; ++t, dst += dstep_b) {
                flat_locate(L, __builtin_amdgcn_readlane(rem_l, t + 1), Ht, rel_n, r_n);
                const uint4 d_n = desc[(t + 1) * KS + rel_n];
                const uint4 re_n = rtab[r_n >> 2];
                const uint32_t w_n = lw_read[8 * (t + 1)];
                put(top_flat_pixels<STRADDLE, NARROW>(C, r, d, w, re.x, re.y, re.z, ((circle_chunks >> t) & 1ull) != 0));
                d = d_n; w = w_n; re = re_n; r = r_n;
            }
            t0 = n_fast;
        }
#pragma unroll 2
        for (int t = t0; t < 64; ++t, dst += dstep_b) {
            const int s_state = __builtin_amdgcn_readlane(state_l, t);
            if (!(s_state & 1)) continue;                                    // wave-uniform: past the end
            int rel, r;
            flat_locate(L, __builtin_amdgcn_readlane(rem_l, t), Ht, rel, r);
            const uint4 d = desc[t * KS + rel];
            const uint4 re = rtab[r >> 2];
            const u32x4 o = top_flat_pixels<STRADDLE, NARROW>(C, r, d, lw_read[8 * t], re.x, re.y, re.z, (s_state & 4) != 0);
            if (s_state & 2) put(o);                                         // every pixel of the chunk is written
            else if ((d.x >> 31) && (((unsigned long long)(base + (uint32_t)t * G)) << 8) + 4u * (unsigned)lane < total_px)
                put(o);                                                      // a chunk at a masked agent's / a run's border, the batch's last chunk
        }
        return false;
    };

    uint32_t base = chunk_begin + g;
    if (base >= chunk_end) return;
    TopFlatPre<K> P = {};
    int state_l = 0, rem_l = 0;
    issue(base, P); flat_wait_loads<0, K>(P); finish(P, state_l, rem_l);
#ifdef RCW_TRACE_WAVES
    int grp = 0;
    unsigned long long t_pause = __builtin_amdgcn_s_memrealtime();           // (a group's "pause": from the end of the previous group's stores to its own first)
#endif
    uint32_t slot = g;                                                       // the slot of the group at `base`
    auto delta = [&](uint32_t sl) -> uint32_t { return G * 64 + R - (sl + R >= G ? G : 0u); };
    while (base + delta(slot) < chunk_end) {                                 // (wave-uniform) there is a next group:
        // its loads go out now, ahead of this group's stores, and are awaited behind them — in one straight line, every
        // iteration, so that no register copy can come between a load and its wait (tools/check_async_loads.py)
        __builtin_amdgcn_wave_barrier();
        issue(base + delta(slot), P);
#ifdef RCW_TRACE_WAVES
        {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && g < 1024 && grp < 18) { g_wave_trace[(g * 20 + grp) * 2] = t_pause; g_wave_trace[(g * 20 + grp) * 2 + 1] = t1; }
            grp += 1;
        }
#endif
        const bool stored_64 = store_group(base, state_l, rem_l);
#ifdef RCW_TRACE_WAVES
        t_pause = __builtin_amdgcn_s_memrealtime();
#endif
        __builtin_amdgcn_wave_barrier();
        if (!stored_64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (fewer than 64 stores behind the loads: wait for everything; names no register)
        flat_wait_loads<63, K>(P);
        finish(P, state_l, rem_l);                                           // ... while this group's stores drain
        base += delta(slot);
        slot = slot + R >= G ? slot + R - G : slot + R;
    }
    __builtin_amdgcn_wave_barrier();
    store_group(base, state_l, rem_l);
#ifdef RCW_TRACE_WAVES
    if (lane == 0 && g < 1024) {
        g_wave_trace[(g * 20 + 19) * 2] = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
        g_wave_trace[(g * 20 + 18) * 2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    }
#endif
}

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------
size_t rcw_step_lds_bytes(const RcwDev& p)
{
    return (((size_t)p.H * p.W + 15) & ~(size_t)15);   // one byte per tile
}
// the cast kernel's LDS: the tile bytes (+ the heading's table slice under the RCW_CAST_TABLE=lds development switch)
static size_t rcw_cast_lds_bytes(const RcwDev& p)                          // guard | tile bytes | guard | re-sampled pose (32 B) [| table slice]
{
    const size_t tiles = (((size_t)p.H * p.W + 2 * (size_t)p.H + 15) & ~(size_t)15);
#ifdef RCW_DEV_SWITCHES
    if (p.cast_table_lds) return tiles + 32 + (size_t)RCW_TABLE_ROWS * p.N * (p.real64 ? 8 : 4);
#endif
    return tiles + 32;
}

// rcw_fill_flat_kernel: the image columns a 256-pixel chunk can touch at this camera height; 0: the kernel does not take it
int rcw_fill_flat_cols(const RcwDev& p)
{
    const int K = 254 / p.Hc + 2;
    return K <= kFlatMaxCols ? K : 0;
}

// which kernel fills the frames of this geometry
enum FillKernel { kFill256, kFillWindow1, kFillWindow2, kFillWindow4, kFillFlat, kFillFrame, kFillAny };
static FillKernel fill_choice(const RcwDev& p, long long total_cols)
{
    if (p.Hc == 256) return kFill256;
    if ((p.Hc & 255) == 0) return kFillWindow1;                             // a 1 KiB chunk is a row block of one column
    if ((p.Hc == 128 || p.Hc == 64) && ((long long)p.N * p.Hc) % 256 == 0 && !p.fill_flat) return p.Hc == 128 ? kFillWindow2 : kFillWindow4;   // 2 / 4 whole columns
    if (rcw_fill_flat_cols(p) && total_cols < (1ll << 31) - 16) return kFillFlat;   // any other height of at least 24 rows
    if (p.N <= 8192 && (long long)p.N * p.Hc < (1ll << 25)) return kFillFrame;
    return kFillAny;
}
const char* rcw_fill_kernel_name(const RcwDev& p, long long total_cols)
{
#ifdef RCW_DEV_SWITCHES
    if (p.step_fused && total_cols == (long long)p.B * p.N && rcw_step_fusable(p)) return "rcw_step256_kernel";
#endif
    // a step that also renders the top view in the fused form: the camera fill of the whole batch and the drawing are ONE launch
    if (p.top_view && p.top_split && p.top_fused && total_cols == (long long)p.B * p.N) return "rcw_fill256_draw_kernel";
    switch (fill_choice(p, total_cols)) {
    case kFill256: return "rcw_fill256_kernel";
    case kFillWindow1: case kFillWindow2: case kFillWindow4: return "rcw_fill_window_kernel";
    case kFillFlat: return "rcw_fill_flat_kernel";
    case kFillFrame: return "rcw_fill_frame_kernel";
    default: return "rcw_fill_any_kernel";
    }
}

hipError_t rcw_launch_fill(const RcwDev& p, const int32_t* col_h, const uint8_t* col_c, uint32_t* frames,
                           long long total_cols, const uint8_t* mask_dev, hipStream_t s)
{
    const int grid = p.fill_grid;
    const long long chunks = total_cols * p.Hc / 256;
    u32x4* const frames4 = reinterpret_cast<u32x4*>(frames);
    switch (fill_choice(p, total_cols)) {
    case kFill256:
#ifdef RCW_DEV_SWITCHES
        if (p.fill_trips >= 0 && !p.fill_plain) {
            switch (p.fill_trips) {
            case 0: hipLaunchKernelGGL(rcw_fill256_trips_kernel<0>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, total_cols, mask_dev); break;
            case 1: hipLaunchKernelGGL(rcw_fill256_trips_kernel<1>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, total_cols, mask_dev); break;
            case 2: hipLaunchKernelGGL(rcw_fill256_trips_kernel<2>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, total_cols, mask_dev); break;
            default: hipLaunchKernelGGL(rcw_fill256_trips_kernel<4>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, total_cols, mask_dev); break;
            }
            break;
        }
#endif
#ifdef RCW_DEV_SWITCHES
        if (p.fill_plain) { hipLaunchKernelGGL(rcw_fill256_kernel<true>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, total_cols, mask_dev); break; }
#endif
        hipLaunchKernelGGL(rcw_fill256_kernel<false>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, total_cols, mask_dev);
        break;
    case kFillWindow1:
        hipLaunchKernelGGL(rcw_fill_window_kernel<1>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, chunks, mask_dev);
        break;
    case kFillWindow2:
        hipLaunchKernelGGL(rcw_fill_window_kernel<2>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, chunks, mask_dev);
        break;
    case kFillWindow4:
        hipLaunchKernelGGL(rcw_fill_window_kernel<4>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames4, chunks, mask_dev);
        break;
    case kFillFlat: {
        // the moving window over 256-pixel chunks of the flat batch
        const int K = rcw_fill_flat_cols(p);
        const size_t lds = (size_t)(kBlock / 64) * 64 * (K + 1) * sizeof(uint4) + 512;   // (+ the fast loop reads a 65th chunk's pairs behind the last wavefront's)
#ifdef RCW_DEV_SWITCHES
#define RCW_FILL_FLAT(AL, KK) do { if (p.fill_pairs == 1) hipLaunchKernelGGL((rcw_fill_flat_kernel<AL, KK, true>), dim3(grid), dim3(2 * kBlock), 2 * lds, s, p, col_h, col_c, frames, total_cols, mask_dev); \
                                   else hipLaunchKernelGGL((rcw_fill_flat_kernel<AL, KK>), dim3(grid), dim3(kBlock), lds, s, p, col_h, col_c, frames, total_cols, mask_dev); } while (0)
#else
#define RCW_FILL_FLAT(AL, KK) hipLaunchKernelGGL((rcw_fill_flat_kernel<AL, KK>), dim3(grid), dim3(kBlock), lds, s, p, col_h, col_c, frames, total_cols, mask_dev)
#endif
#define RCW_FILL_FLAT_K(KK) case KK: if ((p.Hc & 3) == 0) RCW_FILL_FLAT(true, KK); else RCW_FILL_FLAT(false, KK); break
        switch (K) { RCW_FILL_FLAT_K(2); RCW_FILL_FLAT_K(3); RCW_FILL_FLAT_K(4); RCW_FILL_FLAT_K(5); RCW_FILL_FLAT_K(6); RCW_FILL_FLAT_K(7); RCW_FILL_FLAT_K(8);
                     RCW_FILL_FLAT_K(9); RCW_FILL_FLAT_K(11); RCW_FILL_FLAT_K(12);
                     case 10: if ((p.Hc & 3) == 0) return hipErrorInvalidValue; RCW_FILL_FLAT(false, 10); break;   // (29, 30, 31 rows: none a multiple of 4)
                     default: return hipErrorInvalidValue; }
#undef RCW_FILL_FLAT_K
#undef RCW_FILL_FLAT
        break;
    }
    case kFillFrame: {
        const int agents = (int)(total_cols / p.N);
        const size_t lds = (size_t)p.N * 8;
        if ((p.Hc & 3) == 0) hipLaunchKernelGGL(rcw_fill_frame_kernel<true>, dim3(agents), dim3(kBlock), lds, s, p, col_h, col_c, frames, mask_dev);
        else                 hipLaunchKernelGGL(rcw_fill_frame_kernel<false>, dim3(agents), dim3(kBlock), lds, s, p, col_h, col_c, frames, mask_dev);
        break;
    }
    default:
        if ((p.Hc & 3) == 0) hipLaunchKernelGGL(rcw_fill_any_kernel<true>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames, total_cols, mask_dev);
        else                 hipLaunchKernelGGL(rcw_fill_any_kernel<false>, dim3(grid), dim3(kBlock), 0, s, p, col_h, col_c, frames, total_cols, mask_dev);
        break;
    }
    return hipGetLastError();
}

// Dispatch on the compiled-in choices: world-unit type T (p.real64) and the two UNPINNED cast_ray
// switches.  KERNEL is a template name taking <T, TIE_LE, DIST_PRE>.
#define RCW_DISPATCH(KERNEL, GRID, BLOCK, LDS, ...)                                                          \
    do {                                                                                                     \
        if (p.real64) {                                                                                      \
            if (p.tie_le) { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<double, true, true>), GRID, BLOCK, LDS, s, __VA_ARGS__);   \
                            else            hipLaunchKernelGGL((KERNEL<double, true, false>), GRID, BLOCK, LDS, s, __VA_ARGS__); } \
            else          { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<double, false, true>), GRID, BLOCK, LDS, s, __VA_ARGS__);  \
                            else            hipLaunchKernelGGL((KERNEL<double, false, false>), GRID, BLOCK, LDS, s, __VA_ARGS__); } \
        } else {                                                                                             \
            if (p.tie_le) { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<float, true, true>), GRID, BLOCK, LDS, s, __VA_ARGS__);    \
                            else            hipLaunchKernelGGL((KERNEL<float, true, false>), GRID, BLOCK, LDS, s, __VA_ARGS__); }  \
            else          { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<float, false, true>), GRID, BLOCK, LDS, s, __VA_ARGS__);   \
                            else            hipLaunchKernelGGL((KERNEL<float, false, false>), GRID, BLOCK, LDS, s, __VA_ARGS__); } \
        }                                                                                                    \
    } while (0)

hipError_t rcw_launch_cast(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev,
                           hipStream_t s, int first, int count)
{
    if (count < 0) count = p.B - first;
#ifdef RCW_DEV_SWITCHES
    if (p.cast_ballot || p.cast_table_lds || p.cast_r3) {                  // the round-3 kernel and its two rejected variants
        RCW_DISPATCH(rcw_cast_kernel_r3, dim3(count), dim3(p.cast_block), rcw_cast_lds_bytes(p), p, actions_dev, mask_dev, first);
        return hipGetLastError();
    }
#endif
#ifdef RCW_DEV_SWITCHES
    if (p.cast_waves && p.cast_block == 64 && count >= 4 * p.fill_grid) {
        // a wavefront per agent, four agents a workgroup: batches that fill the chip, at most 256 view columns
        const size_t per_agent = (rcw_cast_lds_bytes(p) + 15) & ~(size_t)15;
        RCW_DISPATCH(rcw_cast_waves_kernel, dim3((count + 3) / 4), dim3(kBlock), 4 * per_agent, p, actions_dev, mask_dev, first, first + count, (int)(per_agent / 4));
        return hipGetLastError();
    }
#endif
    RCW_DISPATCH(rcw_cast_kernel, dim3(count), dim3(p.cast_block), rcw_cast_lds_bytes(p), p, actions_dev, mask_dev, first);
    return hipGetLastError();
}

// Same dispatch for a kernel template with a fourth (bool) parameter.
#define RCW_DISPATCH_W(KERNEL, WFLAG, GRID, BLOCK, LDS, ...)                                                 \
    do {                                                                                                     \
        if (p.real64) {                                                                                      \
            if (p.tie_le) { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<double, true, true, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__);   \
                            else            hipLaunchKernelGGL((KERNEL<double, true, false, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__); } \
            else          { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<double, false, true, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__);  \
                            else            hipLaunchKernelGGL((KERNEL<double, false, false, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__); } \
        } else {                                                                                             \
            if (p.tie_le) { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<float, true, true, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__);    \
                            else            hipLaunchKernelGGL((KERNEL<float, true, false, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__); }  \
            else          { if (p.dist_pre) hipLaunchKernelGGL((KERNEL<float, false, true, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__);   \
                            else            hipLaunchKernelGGL((KERNEL<float, false, false, WFLAG>), GRID, BLOCK, LDS, s, __VA_ARGS__); } \
        }                                                                                                    \
    } while (0)

// The one-launch step (rcw_fill256_cast_kernel): the geometries that take it — a 256-row camera view filled by rcw_fill256_kernel's
// window, no top view (its drawing needs the state the same launch commits), a batch of fewer than 2^29 view columns — the bytes of ONE
// of its two slot buffers, and the launch: with_fill = the fill workgroups in front (a step); without, the casting workgroups alone
// (they prime the slots behind a reset / set_state, or for a first step: the camera fill then follows as a launch of its own).
int rcw_step_spec_eligible(const RcwDev& p)
{
    return p.top_view == nullptr && !p.fill_plain && (long long)p.B * p.N < (1ll << 29) && fill_choice(p, (long long)p.B * p.N) == kFill256 ? 1 : 0;
}
size_t rcw_step_spec_slot_bytes(const RcwDev& p) { return (size_t)5 * (size_t)p.B * (size_t)p.N * sizeof(uint16_t); }
hipError_t rcw_launch_step_spec(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev, const uint16_t* slots_in,
                                uint16_t* slots_out, bool with_fill, bool cols, hipStream_t s)
{
    const int icols = cols ? 1 : 0;
    const size_t per_agent = (rcw_cast_lds_bytes(p) + 15) & ~(size_t)15;
    const int fill_blocks = with_fill ? p.fill_grid : 0;
    const long long total_cols = (long long)p.B * p.N;
    u32x4* const out = reinterpret_cast<u32x4*>(p.obs);
    int n_shift = -1;
    for (int k = 0; k < 31; ++k) if (p.N == (1 << k)) n_shift = k;
    const bool wave = p.N <= 64 * kCastCols;                                // a wavefront per agent
    const int cast_blocks = wave ? (p.B + kBlock / 64 - 1) / (kBlock / 64) : p.B;
    const size_t lds = wave ? (kBlock / 64) * per_agent : per_agent;
    const int lds_words = (int)(per_agent / 4);
    if (with_fill) {
        if (wave) RCW_DISPATCH_W(rcw_fill256_cast_kernel, true, dim3(fill_blocks + cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev,
                                 out, total_cols, fill_blocks, slots_in, slots_out, lds_words, n_shift, icols);
        else      RCW_DISPATCH_W(rcw_fill256_cast_kernel, false, dim3(fill_blocks + cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev,
                                 out, total_cols, fill_blocks, slots_in, slots_out, lds_words, n_shift, icols);
    } else {
        if (wave) RCW_DISPATCH_W(rcw_cast_successors_kernel, true, dim3(cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev, slots_out, lds_words);
        else      RCW_DISPATCH_W(rcw_cast_successors_kernel, false, dim3(cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev, slots_out, lds_words);
    }
    return hipGetLastError();
}

#ifdef RCW_DEV_SWITCHES
// the whole step in one launch (rcw_step256_kernel): whether this handle can take it, and the launch
bool rcw_step_fusable(const RcwDev& p)
{
    return p.step_flags != nullptr && p.step_hc != nullptr && p.top_view == nullptr && p.cast_block == 64 && (long long)p.B * p.N < (1ll << 31)
        && fill_choice(p, (long long)p.B * p.N) == kFill256 && !p.fill_plain;
}
hipError_t rcw_launch_step256(const RcwDev& p, const uint8_t* actions_dev, const uint8_t* mask_dev, uint32_t epoch, hipStream_t s)
{
    const size_t per_agent = (rcw_cast_lds_bytes(p) + 15) & ~(size_t)15;
    const int cast_blocks = (p.B + 3) / 4;
    RCW_DISPATCH(rcw_step256_kernel, dim3(p.fill_grid + cast_blocks), dim3(kBlock), 4 * per_agent, p, actions_dev, mask_dev,
                 reinterpret_cast<u32x4*>(p.obs), (long long)p.B * p.N, p.fill_grid, (int)(per_agent / 4), epoch);
    return hipGetLastError();
}
#endif

size_t rcw_top_view_lds_bytes(const RcwDev& p)
{
    return 16 + (size_t)(p.top_lds > 0 ? p.top_lds : 1) * 4 * top_buf_words(p);      // counters + the ring of p.top_lds buffers
}

hipError_t rcw_launch_top_view(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s)
{
    if (p.top_lds) {
        const int grid = p.B < p.top_grid ? p.B : p.top_grid;              // persistent: 4 workgroups of 8 wavefronts per CU
        RCW_DISPATCH(rcw_top_view_kernel, dim3(grid), dim3(kTopBlock), rcw_top_view_lds_bytes(p), p, mask_dev);
    } else {
        RCW_DISPATCH(rcw_top_view_inplace_kernel, dim3(p.B), dim3(kBlock), rcw_step_lds_bytes(p), p, mask_dev);
    }
    return hipGetLastError();
}

// The two-kernel top view (see rcw_top_draw_kernel): whether this geometry takes it
// ... as the number of rows of a unit (256: rcw_top_store_kernel; 128, 64 or 32: rcw_top_store_units_kernel), 0: not taken.
// A unit is a run of rows of ONE image column that holds whole tiles, a lane's four pixels a whole quarter of one.
int rcw_top_split_unit(const RcwDev& p)
{
    const long long Ht = (long long)p.H * p.pu, Wt = (long long)p.W * p.pu;
    if (p.pu < 8 || 2 * p.top_rp > 31 || p.N > 4096) return 0;                                   // (the draw kernel ranks a line within its length class in 12 bits)
    int unit = 0;
    if (256 % p.pu == 0 && Ht % 256 == 0) unit = 256;
#ifdef RCW_DEV_SWITCHES
    else if (128 % p.pu == 0 && Ht % 128 == 0 && 128 / p.pu <= 14) unit = 128;   // (every such geometry takes the flat kernel: RCW_TOP_FLAT=0 only)
#endif
    else if (64 % p.pu == 0 && Ht % 64 == 0) unit = 64;
    else if (32 % p.pu == 0 && Ht % 32 == 0) unit = 32;
    if (!unit) return 0;
    if ((long long)p.B * Wt * (Ht / unit) + 8ll * 64 * p.top_store_grid * (kBlock / 64) >= (1ll << 31)) return 0;   // unit / chunk ids in 32 bits
    if ((long long)p.B * Wt * (Ht >> 5) >= (1ll << 31) - 64) return 0;                                               // plane word offsets
    return 4 * top_draw_lds_words(p) <= 159 * 1024 ? unit : 0;                                       // the draw kernel's LDS: plane + ray lists
}
// rcw_top_store_flat_kernel: the image columns a 256-pixel chunk can touch in this geometry; 0: the kernel does not take it
static size_t top_circle_table_bytes(const RcwDev& p) { return (size_t)(p.top_rp + 1) * ((2 * p.top_rp + 1 + 31) / 32 + 2) * 4; }
static size_t top_flat_plane_words(const RcwDev& p) { return (((size_t)p.H * p.pu * p.W * p.pu + 255 + 255) / 256) * 8; }
static size_t top_store_flat_lds_bytes(const RcwDev& p, int K)             // plane words | descriptors | circle rows | row table
{
    return (size_t)(kBlock / 64) * 512 * 4 + (size_t)(kBlock / 64) * 64 * K * 16 + ((top_circle_table_bytes(p) + 15) & ~(size_t)15) + (size_t)4 * p.H * p.pu;
}
int rcw_top_flat_cols(const RcwDev& p)
{
    const long long Ht = (long long)p.H * p.pu, Wt = (long long)p.W * p.pu;
    if (p.pu < 9 || (Ht & 3) != 0 || Ht > 16384 || Wt > 16384 || p.H > 65535 || p.top_rp > 8191 || p.N > 4096) return 0;
    const int K = (int)(251 / Ht) + 2;
    if (K > 7) return 0;                                                              // (the store kernel is instantiated for 2..7)
    if (top_circle_table_bytes(p) > 16 * 1024) return 0;
    if (top_store_flat_lds_bytes(p, K) > 64 * 1024) return 0;                                 // (plane words, descriptors, circle rows, row table: the default limit is kept)
    if (4 * top_draw_lds_words(p) > 159 * 1024) return 0;                                     // the draw kernel's LDS: plane + ray lists
    const long long chunks = ((long long)p.B * Ht * Wt + 255) / 256;
    if ((long long)p.B * Wt >= (1ll << 31) - 64) return 0;                                    // image columns of the flat batch in 32 bits
    if (chunks + 64ll * p.top_store_grid * (kBlock / 64) >= (1ll << 31)) return 0;            // chunk ids
    if ((long long)p.B * (long long)top_flat_plane_words(p) >= (1ll << 31) - 64) return 0;    // plane word offsets
    return K;
}
size_t rcw_top_plane_bytes(const RcwDev& p)
{
    if (p.top_flat) return (size_t)p.B * top_flat_plane_words(p) * 4 + 64;
    return (size_t)p.B * p.W * p.pu * ((size_t)p.H * p.pu / 32) * 4 + 64;                     // (+ a short last chunk's reach)
}
int32_t rcw_top_plane_words(const RcwDev& p) { return (int32_t)top_flat_plane_words(p); }
size_t rcw_top_codes_bytes(const RcwDev& p)
{
    if (p.top_flat) return 64;                                              // (the flat store kernel reads tile_map itself)
    return (size_t)p.B * p.W * ((size_t)p.H * p.pu / p.top_unit_px) * sizeof(uint2);
}

// agents [first, first + count): the draw kernel's workgroups / the store kernel's chunks of that run (an image is a
// whole number of 1 KiB chunks in every geometry rcw_top_split_unit takes)
hipError_t rcw_launch_top_draw(const RcwDev& p, const uint8_t* mask_dev, int first, int count, hipStream_t s, int block)
{
    RCW_DISPATCH(rcw_top_draw_kernel, dim3(count * (p.top_parts > 1 ? p.top_parts : 1)), dim3(block > 0 ? block : p.top_draw_block), 4 * top_draw_lds_words(p), p, mask_dev, first);
    return hipGetLastError();
}
// the camera fill of the whole batch + the drawing of every agent in one launch (rcw_fill256_draw_kernel): whether this handle's
// geometry takes it, and the launch
int rcw_fill_draw_fusable(const RcwDev& p)
{
    return p.top_split && p.top_runs <= 1 && p.top_draw_block == kBlock && !p.fill_plain && 4 * top_draw_lds_words(p) <= 64 * 1024 &&
           fill_choice(p, (long long)p.B * p.N) == kFill256 && (long long)p.fill_grid + p.B < (1ll << 31);
}
hipError_t rcw_launch_fill256_draw(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s)
{
    u32x4* const frames4 = reinterpret_cast<u32x4*>(p.obs);
    RCW_DISPATCH(rcw_fill256_draw_kernel, dim3(p.fill_grid + p.B), dim3(kBlock), 4 * top_draw_lds_words(p), p, p.col_h, p.col_c, frames4,
                 (long long)p.B * p.N, mask_dev, p.fill_grid, 0);
    return hipGetLastError();
}
// The store kernel may FOLLOW the draw kernel (top_follow_wait) only where a draw workgroup still finds room on a CU whose store
// workgroups — resident for the whole launch, and waiting — are already there (and, inside a step, the camera fill's): wavefronts
// (32 a CU; 28 counted, what the draw kernel was seen to reach) and LDS (160 KiB).
// draw workgroups that fit on a CU together (LDS, wavefronts): what one "round" of the draw kernel is
int rcw_top_draw_per_cu(const RcwDev& p, int draw_block)
{
    const size_t lds = 4 * top_draw_lds_words(p);
    int n = lds ? (int)((size_t)(160 * 1024) / lds) : 8;
    const int by_waves = 28 / (draw_block / 64 > 0 ? draw_block / 64 : 1);
    if (n > by_waves) n = by_waves;
    return n < 1 ? 1 : n;
}
int rcw_top_follow_fits(const RcwDev& p, int draw_block, bool beside_fill, int cus)
{
    if (!p.top_split || cus <= 0) return 0;
    const int store_wgs = (p.top_store_grid + cus - 1) / cus;               // per CU
    size_t store_lds = (kBlock / 64) * 512 * 4;                              // rcw_top_store_kernel's plane words
    if (p.top_flat) store_lds = top_store_flat_lds_bytes(p, p.top_flat);
    else if (p.top_unit_px != 256) store_lds = (size_t)(kBlock / 64) * (512 + 3 * 64 * (256 / p.top_unit_px)) * 4;
    const int fill_wgs = beside_fill ? (p.fill_grid + cus - 1) / cus : 0;
    const int waves = draw_block / 64 + (store_wgs + fill_wgs) * (kBlock / 64);
    const size_t lds = 4 * top_draw_lds_words(p) + store_wgs * (store_lds + 512) + fill_wgs * (size_t)(8 * 1024);
    return waves <= 28 && lds <= 156 * 1024 ? 1 : 0;
}
hipError_t rcw_launch_top_store(const RcwDev& p, const uint8_t* mask_dev, int first, int count, hipStream_t s)
{
    const dim3 grid(p.top_store_grid), block(kBlock);
    if (p.top_flat) {
        // the chunks of the flat batch that hold a pixel of agents [first, first + count)
        const unsigned long long px = (unsigned long long)p.H * p.pu * p.W * p.pu;
        const uint32_t c0 = (uint32_t)((px * (unsigned)first) >> 8), c1 = (uint32_t)((px * (unsigned)(first + count) + 255) >> 8);
        const size_t lds = top_store_flat_lds_bytes(p, p.top_flat);
        const bool straddle = (p.pu & 3) != 0;
        const bool narrow = p.pu >= 19;                                      // 255 / pu + 2 <= 16 tiles in a 256-row run: one word of codes
#define RCW_FLAT(ST, NA, KK) hipLaunchKernelGGL((rcw_top_store_flat_kernel<ST, NA, KK>), grid, block, lds, s, p, mask_dev, c0, c1, first, first + count)
#define RCW_FLAT_K(KK) case KK: if (straddle) { if (narrow) RCW_FLAT(true, true, KK); else RCW_FLAT(true, false, KK); } \
                                else          { if (narrow) RCW_FLAT(false, true, KK); else RCW_FLAT(false, false, KK); } break
        // (7 columns a chunk are images of 44 or 48 rows: at 19 pixels a tile and more that would be a map of two rows — there is no such instantiation;
        // and 6 columns a chunk — 52, 56 or 60 rows — at such a scale are three tile rows of 20 pixels: a multiple of 4, no straddling)
        switch (p.top_flat) { RCW_FLAT_K(2); RCW_FLAT_K(3); RCW_FLAT_K(4); RCW_FLAT_K(5);
                              case 6: if (straddle && narrow) return hipErrorInvalidValue;
                                      if (straddle) RCW_FLAT(true, false, 6); else if (narrow) RCW_FLAT(false, true, 6); else RCW_FLAT(false, false, 6); break;
                              case 7: if (narrow) return hipErrorInvalidValue; if (straddle) RCW_FLAT(true, false, 7); else RCW_FLAT(false, false, 7); break;
                              default: return hipErrorInvalidValue; }
#undef RCW_FLAT_K
#undef RCW_FLAT
        return hipGetLastError();
    }
    const uint32_t per_agent = (uint32_t)(((long long)p.H * p.pu * p.W * p.pu) >> 8);
    const uint32_t c0 = (uint32_t)first * per_agent, c1 = (uint32_t)(first + count) * per_agent;
    // (plain instead of non-temporal stores — p.top_store_plain — and units of 128 rows, which the flat kernel has taken over, are
    // choices of the development build only: the shipped library carries no instantiation it cannot reach)
#ifdef RCW_DEV_SWITCHES
#define RCW_STORE(KERNEL, ...) do { if (p.top_store_plain) hipLaunchKernelGGL((KERNEL<true, __VA_ARGS__>), grid, block, 0, s, p, mask_dev, c0, c1); \
                                    else hipLaunchKernelGGL((KERNEL<false, __VA_ARGS__>), grid, block, 0, s, p, mask_dev, c0, c1); } while (0)
#else
#define RCW_STORE(KERNEL, ...) hipLaunchKernelGGL((KERNEL<false, __VA_ARGS__>), grid, block, 0, s, p, mask_dev, c0, c1)
#endif
    if (p.top_unit_px == 128) {
#ifdef RCW_DEV_SWITCHES
        RCW_STORE(rcw_top_store_units_kernel, 2);
#else
        return hipErrorInvalidValue;
#endif
    }
    else if (p.top_unit_px == 64) RCW_STORE(rcw_top_store_units_kernel, 4);
    else if (p.top_unit_px == 32) RCW_STORE(rcw_top_store_units_kernel, 8);
    else if (p.pu < 16) RCW_STORE(rcw_top_store_kernel, true);               // 32 tiles in a chunk
    else RCW_STORE(rcw_top_store_kernel, false);
#undef RCW_STORE
    return hipGetLastError();
}

// Above 64 KiB of dynamic LDS a kernel has to be told so once (the CU has 160 KiB).  The attribute belongs to the
// FUNCTION, i.e. to every handle on the device: it is set once per device, to the fixed cap the geometry selection
// works with (16 B + 156 KiB for the ring kernel, 156 KiB for the draw kernel), never to one handle's own need — a
// second handle with a smaller image must not lower the limit under a first one's feet.
hipError_t rcw_prepare_top_view(const RcwDev& p, int device)
{
    static std::mutex mu;
    static bool done[64] = {};
    if (!p.top_view) return hipSuccess;
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && device < 64 && done[device]) return hipSuccess;
    const int cap = 160 * 1024;
    hipError_t e = hipSuccess;
#define RCW_TOP_ATTR(TT, A, B_)                                                                                         \
    if (e == hipSuccess)                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rcw_top_view_kernel<TT, A, B_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, cap);                                      \
    if (e == hipSuccess)                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rcw_top_draw_kernel<TT, A, B_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, cap)
    RCW_TOP_ATTR(double, false, false); RCW_TOP_ATTR(double, false, true); RCW_TOP_ATTR(double, true, false); RCW_TOP_ATTR(double, true, true);
    RCW_TOP_ATTR(float, false, false); RCW_TOP_ATTR(float, false, true); RCW_TOP_ATTR(float, true, false); RCW_TOP_ATTR(float, true, true);
#undef RCW_TOP_ATTR
    if (e == hipSuccess && device >= 0 && device < 64) done[device] = true;
    return e;
}

hipError_t rcw_launch_reset(const RcwDev& p, const uint8_t* mask_dev, hipStream_t s)
{
    if (p.real64) hipLaunchKernelGGL(rcw_reset_kernel<double>, dim3((p.B + 63) / 64), dim3(64), 0, s, p, mask_dev);
    else          hipLaunchKernelGGL(rcw_reset_kernel<float>, dim3((p.B + 63) / 64), dim3(64), 0, s, p, mask_dev);
    return hipGetLastError();
}

// pos: float2* for a Float32 world, double2* for a Float64 world
hipError_t rcw_launch_set_state(const RcwDev& p, const int2* goal, const void* pos,
                                const int32_t* dir, const uint8_t* mask_dev, hipStream_t s)
{
    if (p.real64)
        hipLaunchKernelGGL(rcw_set_state_kernel<double>, dim3((p.B + 63) / 64), dim3(64), 0, s, p, goal,
                           static_cast<const double2*>(pos), dir, mask_dev);
    else
        hipLaunchKernelGGL(rcw_set_state_kernel<float>, dim3((p.B + 63) / 64), dim3(64), 0, s, p, goal,
                           static_cast<const float2*>(pos), dir, mask_dev);
    return hipGetLastError();
}

hipError_t rcw_launch_init_tile_map(const RcwDev& p, hipStream_t s)
{
    hipLaunchKernelGGL(rcw_init_tile_map_kernel, dim3((p.B + 63) / 64), dim3(64), 0, s, p);
    return hipGetLastError();
}

hipError_t rcw_launch_rays(const RcwDev& p, int32_t first, int32_t count, RcwRayOut out, hipStream_t s)
{
    RCW_DISPATCH(rcw_rays_kernel, dim3(count), dim3(kBlock), rcw_step_lds_bytes(p), p, first, out);
    return hipGetLastError();
}
#undef RCW_DISPATCH

hipError_t rcw_launch_expand(const RcwDev& p, const int32_t* col_h, const uint8_t* col_c,
                             int32_t count, uint32_t* frames, hipStream_t s)
{
    return rcw_launch_fill(p, col_h, col_c, frames, (long long)count * p.N, nullptr, s);
}
