#pragma once
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



// (every translation unit includes this header: the helpers below are internal to each — inlined into its kernels)
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
#include "dev/cast_ray_ballot.inc"   // RCW_CAST_MARCH=ballot, the ballot-bounded march (measured, rejected)
#endif

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

// ---- loads shared by the casting and the drawing kernels ----------------------------------------------------------------------
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

constexpr int kCastCols = 4;    // view columns a lane keeps in registers (cast_block is chosen so that a lane has at most 4)
constexpr int kCastTiles = 4;   // tiles a lane unpacks from words requested in batch 1 (a 32x32 map on 256 lanes); larger maps loop

// ---- the moving window (camera fill, top view store): what its kernels share -----------------------------------------------------
template <bool PLAIN>
__device__ __forceinline__ void store16(u32x4* dst, u32x4 v)
{
    if (PLAIN) *dst = v; else __builtin_nontemporal_store(v, dst);
}

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

}  // namespace

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
