/*
 * rcw_oracle.c — CPU restatement of the RayCastWorlds.jl SingleRoom step/render path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / the reported CPU baseline.  The product (librcw_hip.so) never links, loads
 * or falls back to anything in oracle/.
 *
 * PARITY UNPINNED.  The reference is Julia and there is no Julia toolchain here, and its
 * tests hold no golden vectors for this path (test/runtests.jl:15-44 checks invariants
 * only).  Three pieces of arithmetic live in un-vendored dependencies and are restated
 * from their published algorithms: RayCaster.cast_ray (RayCaster 0.1.x, call site
 * src/single_room.jl:223), StaticArrays 1.2 `normalize` (SR:221) and Julia Base
 * `LinRange` indexing (SR:218,221).  Each such choice is marked UNPINNED below and is
 * switchable through rcw_config (include/rcw.h).  What pins this file is (1) the
 * hand-derived known-answer vectors in tests/golden/ (derived from the reference text,
 * SURVEY.md §8c) and (2) an independent second restatement in oracle/pyref.py.
 *
 * Every function cites the reference lines it follows (SR = src/single_room.jl,
 * CD = src/collision_detection.jl, UT = src/utils.jl).  All arithmetic is Float32 with
 * one IEEE rounding per operation and no fused multiply-add (build with
 * -ffp-contract=off), except the Float64 lerp of LinRange.
 *
 * Layouts are the reference's column-major ones with a trailing batch axis
 * (include/rcw.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rcw.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EXPORT __attribute__((visibility("default")))

/* World-unit type T of the reference (SR:259).  This file is compiled twice: as is (Float32,
 * librcw_oracle.so) and with -DORC_REAL64 (Float64, librcw_oracle64.so); every arithmetic
 * operation of the path is the same operation in T.  R (the reward type, SR:266) is independent of T:
 * cfg.reward_type, orc_set_reward below. */
#ifdef ORC_REAL64
typedef double real;
#define RC(x) x
#define R_FLOOR floor
#define R_FABS fabs
#define R_SQRT sqrt
#define CFG_RADIUS(c) ((c)->player_radius_wu_f64)
#define CFG_INC(c) ((c)->position_increment_wu_f64)
#define CFG_FOV(c) ((c)->semi_field_of_view_wu_f64)
#define CFG_CAMH(c) ((c)->camera_height_tile_wu_f64)
#else
typedef float real;
#define RC(x) x##f
#define R_FLOOR floorf
#define R_FABS fabsf
#define R_SQRT sqrtf
#define CFG_RADIUS(c) ((c)->player_radius_wu)
#define CFG_INC(c) ((c)->position_increment_wu)
#define CFG_FOV(c) ((c)->semi_field_of_view_wu)
#define CFG_CAMH(c) ((c)->camera_height_tile_wu)
#endif

/* ------------------------------------------------------------------------------------
 * Counter-based generator used by reset (the build's own; Julia's streams are not
 * reproducible).  Restated independently in csrc/rcw_rng.h — the spec is
 * DESIGN.md "Reset generator".
 * ---------------------------------------------------------------------------------- */
static uint64_t orc_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
static uint64_t orc_episode_key(uint64_t seed, uint64_t agent, uint64_t episode)
{
    uint64_t k = orc_mix64(seed + 0x9e3779b97f4a7c15ULL * (agent + 1));
    return orc_mix64(k ^ (episode * 0xd1b54a32d192ed03ULL));
}
static uint64_t orc_draw(uint64_t key, uint64_t n)
{
    return orc_mix64(key + 0x9e3779b97f4a7c15ULL * (n + 1));
}
/* uniform integer on 0..range-1: high 64 bits of u*range */
static uint64_t orc_below(uint64_t u, uint64_t range)
{
    return (uint64_t)(((__uint128_t)u * (__uint128_t)range) >> 64);
}

/* ------------------------------------------------------------------------------------
 * A.1 direction table  (SR:65-69)
 * ---------------------------------------------------------------------------------- */
ORC_EXPORT void orc_direction_table(int32_t nd, real* out /* (2, nd) */)
{
    for (int32_t i = 1; i <= nd; ++i) {
        /* theta_wu = (i - 1) * 2 * pi / num_directions, left to right, Float64 (SR:67) */
        double theta = (double)((int64_t)(i - 1) * 2) * 3.141592653589793 / (double)nd;
        out[2 * (i - 1) + 0] = (real)cos(theta);   /* convert(T, cos(theta)) SR:68 */
        out[2 * (i - 1) + 1] = (real)sin(theta);
    }
}

/* ------------------------------------------------------------------------------------
 * A.3 ray fan for one heading  (SR:193, SR:214-221)
 * ---------------------------------------------------------------------------------- */
ORC_EXPORT void orc_ray_fan(const rcw_config* cfg, const real* dir /* (2) */,
                            real* rays /* (2, N) */)
{
    const int32_t N = cfg->num_rays;
    const real fov = CFG_FOV(cfg);
    const real d1 = dir[0], d2 = dir[1];
    /* rotate_minus_90(vec) = (vec[2], -vec[1])  SR:193,215 */
    const real c1 = d2, c2 = -d1;
    /* first = dir + fov * cam ; last = dir - fov * cam   SR:216-217 */
    const real f1 = d1 + fov * c1, f2 = d2 + fov * c2;
    const real l1 = d1 - fov * c1, l2 = d2 - fov * c2;
    /* UNPINNED (Julia Base range.jl): range(first, last, length=N) on SVector is a
     * LinRange; element i is lerpi(i-1, max(N-1,1), first, last) =
     * T((1-t)*first + t*last) with t = (i-1)/lendiv in Float64.  SR:218,221 */
    const int32_t lendiv = (N - 1 > 1) ? N - 1 : 1;
    for (int32_t i = 1; i <= N; ++i) {
        const double t = (double)(i - 1) / (double)lendiv;
        const real u1 = (real)((1.0 - t) * (double)f1 + t * (double)l1);
        const real u2 = (real)((1.0 - t) * (double)f2 + t * (double)l2);
        /* UNPINNED (StaticArrays): normalize(a) = inv(norm(a)) * a, norm = sqrt(sum abs2) */
        const real n = R_SQRT(u1 * u1 + u2 * u2);
        real r1, r2;
        if (cfg->normalize_mode == RCW_NORMALIZE_DIVIDE) {
            r1 = u1 / n; r2 = u2 / n;
        } else {
            const real inv = RC(1.0) / n;
            r1 = inv * u1; r2 = inv * u2;
        }
        rays[2 * (i - 1) + 0] = r1;
        rays[2 * (i - 1) + 1] = r2;
    }
}

/* ------------------------------------------------------------------------------------
 * A.4 RayCaster.cast_ray  (external package; call site SR:223).  UNPINNED: canonical
 * grid DDA.  obst is Bool (H, W) column-major, 1 byte per tile.  Returns 0, or
 * RCW_ERR_OUT_OF_BOUNDS where Julia would raise BoundsError on obstacle_map[i, j].
 * ---------------------------------------------------------------------------------- */
ORC_EXPORT int orc_cast_ray(const uint8_t* obst, int32_t H, int32_t W, real x, real y,
                            real dx, real dy, int32_t tie_break, int32_t dist_mode,
                            int64_t* i_hit, int64_t* j_hit, int64_t* hit_dim, real* dist)
{
    int64_t i = (int64_t)R_FLOOR(x) + 1;   /* wu_to_tu UT:5 */
    int64_t j = (int64_t)R_FLOOR(y) + 1;
    const real ddx = R_FABS(RC(1.0) / dx);
    const real ddy = R_FABS(RC(1.0) / dy);
    int64_t si, sj;
    real sx, sy;
    if (dx < RC(0.0)) { si = -1; sx = (x - (real)(i - 1)) * ddx; }
    else           { si = +1; sx = ((real)i - x) * ddx; }
    if (dy < RC(0.0)) { sj = -1; sy = (y - (real)(j - 1)) * ddy; }
    else           { sj = +1; sy = ((real)j - y) * ddy; }
    int64_t dim = 0;
    real d = RC(0.0);
    for (;;) {
        if (i < 1 || i > H || j < 1 || j > W) return RCW_ERR_OUT_OF_BOUNDS;
        if (obst[(i - 1) + (int64_t)H * (j - 1)]) break;
        const int x_first = (tie_break == RCW_DDA_TIE_X_FIRST_ON_LE) ? (sx <= sy) : (sx < sy);
        if (x_first) { d = sx; sx = sx + ddx; i += si; dim = 1; }
        else         { d = sy; sy = sy + ddy; j += sj; dim = 2; }
    }
    if (dist_mode == RCW_DDA_DIST_SIDE_MINUS_DELTA) {
        if (dim == 1) d = sx - ddx;
        else if (dim == 2) d = sy - ddy;
    }
    *i_hit = i; *j_hit = j; *hit_dim = dim; *dist = d;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A.2 is_player_colliding  (CD:1-42).  layer is Bool (H, W) column-major.
 * Returns 0/1, or RCW_ERR_OUT_OF_BOUNDS for Julia's BoundsError at CD:35.
 * ---------------------------------------------------------------------------------- */
ORC_EXPORT int orc_is_player_colliding(const uint8_t* layer, int32_t H, int32_t W,
                                       real px, real py, real radius, int32_t oob_empty)
{
    const real half = RC(0.5);                           /* StdSquare(0.5) CD:24 */
    const int64_t it = (int64_t)R_FLOOR(px) + 1;        /* wu_to_tu CD:27-28, UT:5 */
    const int64_t jt = (int64_t)R_FLOOR(py) + 1;
    for (int64_t j = jt - 1; j <= jt + 1; ++j) {       /* CD:30 */
        for (int64_t i = it - 1; i <= it + 1; ++i) {   /* CD:31 */
            const real cx = (real)i - half;          /* CD:33-34 */
            const real cy = (real)j - half;
            if (i < 1 || i > H || j < 1 || j > W) {
                if (oob_empty) continue;   /* RCW_OOB_TREAT_EMPTY (not the reference) */
                return RCW_ERR_OUT_OF_BOUNDS;
            }
            if (layer[(i - 1) + (int64_t)H * (j - 1)]) {   /* && short-circuit CD:35 */
                const real qx = px - cx, qy = py - cy;    /* position .- tile CD:35 */
                /* get_projection: clamp.(q, -h, h) CD:9-12 */
                const real sx = qx < -half ? -half : (qx > half ? half : qx);
                const real sy = qy < -half ? -half : (qy > half ? half : qy);
                const real vx = qx - sx, vy = qy - sy;    /* CD:16 */
                if (vx * vx + vy * vy < radius * radius) return 1;   /* CD:18 */
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Batched world: B independent SingleRoomWorld + SingleRoom (SR:21-40, SR:241-256)
 * ---------------------------------------------------------------------------------- */
typedef struct orc_batch {
    rcw_config cfg;
    int32_t B, H, W, N, nd, Hc, nchunks;
    uint64_t seed;
    real* directions;      /* (2, nd) */
    real* ray_table;       /* (2, N, nd) normalized ray directions per heading */
    uint8_t* wall;          /* (H, W, B) Bool layer WALL */
    uint8_t* goalmap;       /* (H, W, B) Bool layer GOAL */
    real* pos;             /* (2, B) */
    int32_t* dir;           /* (B) */
    int32_t* goal;          /* (2, B) 1-based */
    void* reward;           /* (B) of R = cfg.reward_type (SR:33) */
    uint8_t* done;          /* (B) */
    uint32_t* episode;      /* (B) */
    int32_t* status;        /* (B) sticky per-agent error */
    /* ray buffers SR:29-31,39 */
    int64_t* ray_stop;      /* (2, N, B) */
    int64_t* ray_dim;       /* (N, B) */
    real* ray_dist;        /* (N, B) */
    real* ray_dirs;        /* (2, N, B) */
    /* camera view + compact descriptors */
    uint32_t* camera_view;  /* (Hc, N, B) */
    int32_t* col_height;    /* (N, B) indexed by image column k */
    uint8_t* col_colour;    /* (N, B) */
    int render;             /* 0: skip the pixel fill (descriptor only) */
    uint32_t* top_view;     /* (H*pu, W*pu, B) when cfg.render_top_view */
    int32_t Ht, Wt;
} orc_batch;

ORC_EXPORT void orc_destroy(orc_batch* b)
{
    if (!b) return;
    free(b->directions); free(b->ray_table); free(b->wall); free(b->goalmap); free(b->pos);
    free(b->dir); free(b->goal); free(b->reward); free(b->done); free(b->episode);
    free(b->status); free(b->ray_stop); free(b->ray_dim); free(b->ray_dist);
    free(b->ray_dirs); free(b->camera_view); free(b->top_view); free(b->col_height); free(b->col_colour);
    free(b);
}

static void orc_build_ray_table(orc_batch* b)
{
    for (int32_t d = 0; d < b->nd; ++d)
        orc_ray_fan(&b->cfg, b->directions + 2 * d, b->ray_table + (size_t)2 * b->N * d);
}

/* cast_rays!(world)  SR:195-231 for agent a */
static void orc_cast_rays_agent(orc_batch* b, int32_t a)
{
    const int32_t H = b->H, W = b->W, N = b->N;
    const size_t HW = (size_t)H * W;
    /* obstacle_map = any(tile_map, dims = 1)  SR:209 */
    uint8_t obst_stack[1024];
    uint8_t* obst = HW <= sizeof obst_stack ? obst_stack : (uint8_t*)malloc(HW);
    for (size_t t = 0; t < HW; ++t) obst[t] = b->wall[HW * a + t] | b->goalmap[HW * a + t];
    const real x = b->pos[2 * a], y = b->pos[2 * a + 1];
    const real* fan = b->ray_table + (size_t)2 * N * b->dir[a];   /* SR:214-221 */
    for (int32_t i = 0; i < N; ++i) {                               /* SR:220 */
        const real dx = fan[2 * i], dy = fan[2 * i + 1];
        int64_t ih = 0, jh = 0, dim = 0; real dist = RC(0.0);
        int rc = orc_cast_ray(obst, H, W, x, y, dx, dy, b->cfg.dda_tie_break,
                              b->cfg.dda_distance, &ih, &jh, &dim, &dist);   /* SR:223 */
        if (rc != 0) { b->status[a] = rc; ih = 1; jh = 1; dim = 0; dist = RC(0.0); }
        const size_t r = (size_t)N * a + i;
        b->ray_dirs[2 * r] = dx; b->ray_dirs[2 * r + 1] = dy;       /* SR:222 */
        b->ray_stop[2 * r] = ih; b->ray_stop[2 * r + 1] = jh;       /* SR:224-225 */
        b->ray_dim[r] = dim;                                        /* SR:226 */
        b->ray_dist[r] = dist;                                      /* SR:227 */
    }
    if (obst != obst_stack) free(obst);
}

/* update_camera_view!(env)  SR:374-444 for agent a */
static void orc_update_camera_view_agent(orc_batch* b, int32_t a)
{
    const int32_t H = b->H, N = b->N, Hc = b->Hc;
    const size_t HW = (size_t)H * b->W;
    const rcw_config* c = &b->cfg;
    const real p1 = b->directions[2 * b->dir[a]], p2 = b->directions[2 * b->dir[a] + 1]; /* SR:400 */
    uint32_t* view = b->camera_view + (size_t)Hc * N * a;
    for (int32_t i = 1; i <= N; ++i) {                              /* SR:401 */
        const size_t r = (size_t)N * a + (i - 1);
        const real r1 = b->ray_dirs[2 * r], r2 = b->ray_dirs[2 * r + 1];
        /* projected = dist * sum(dir .* ray)  SR:404 */
        const real dot = p1 * r1 + p2 * r2;
        const real projected = b->ray_dist[r] * dot;
        /* height_line = camh * num_rays / (2 * fov * projected)  SR:406 (left assoc.) */
        const real num = CFG_CAMH(c) * (real)N;
        const real den = (RC(2.0) * CFG_FOV(c)) * projected;
        const real height_line = num / den;
        int64_t h;
        if (isfinite(height_line)) {                                /* SR:407-411 */
            const real fl = R_FLOOR(height_line);
            /* floor(Int, x): Julia raises InexactError outside Int64; saturate instead */
            if (fl >= RC(9.2233720368547758e18)) h = INT64_MAX;
            else if (fl <= -RC(9.2233720368547758e18)) h = INT64_MIN;
            else h = (int64_t)fl;
        } else {
            h = Hc;
        }
        const int64_t dim = b->ray_dim[r];
        const int64_t ih = b->ray_stop[2 * r], jh = b->ray_stop[2 * r + 1];
        const int is_wall = b->wall[HW * a + (size_t)(ih - 1) + (size_t)H * (jh - 1)]; /* SR:417 */
        uint32_t colour; uint8_t cid;
        if (is_wall) {                                              /* SR:417-423 */
            if (dim == 1) { colour = c->wall_dim_1_color; cid = RCW_COLOUR_WALL_DIM_1; }
            else          { colour = c->wall_dim_2_color; cid = RCW_COLOUR_WALL_DIM_2; }
        } else {                                                    /* SR:424-429 */
            if (dim == 1) { colour = c->goal_dim_1_color; cid = RCW_COLOUR_GOAL_DIM_1; }
            else          { colour = c->goal_dim_2_color; cid = RCW_COLOUR_GOAL_DIM_2; }
        }
        const int32_t k = N - i + 1;                                /* SR:431 */
        b->col_height[(size_t)N * a + (k - 1)] =
            h > INT32_MAX ? INT32_MAX : (h < INT32_MIN ? INT32_MIN : (int32_t)h);
        b->col_colour[(size_t)N * a + (k - 1)] = cid;
        if (!b->render) continue;
        uint32_t* col = view + (size_t)Hc * (k - 1);
        if (h >= Hc - 1) {                                          /* SR:433-434 */
            for (int32_t q = 0; q < Hc; ++q) col[q] = colour;
        } else {
            const int64_t pad = (Hc - h) / 2;                       /* SR:436 */
            /* 1-based rows 1:pad, pad+1:Hc-pad, Hc-pad+1:Hc  SR:437-439; a negative h
             * (unreachable: distances are positive) would make Julia throw BoundsError,
             * clamp to the frame instead */
            const int64_t top = pad > Hc ? Hc : pad;
            const int64_t bot = Hc - pad < 0 ? 0 : Hc - pad;
            for (int64_t q = 0; q < top; ++q) col[q] = c->ceiling_color;
            for (int64_t q = top; q < bot; ++q) col[q] = colour;
            for (int64_t q = bot > top ? bot : top; q < Hc; ++q) col[q] = c->floor_color;
        }
    }
}

/* ---- SimpleDraw 0.3 primitives used by the top view (un-vendored: UNPINNED) ------------------
 * image is column-major (Ht, Wt); (i, j) are 1-based; pixels off the image are skipped. */
static void sd_put_pixel(uint32_t* img, int64_t Ht, int64_t Wt, int64_t i, int64_t j, uint32_t c)
{
    if (i >= 1 && i <= Ht && j >= 1 && j <= Wt) img[(i - 1) + Ht * (j - 1)] = c;
}
/* SD.Line: ASSUMED Bresenham, all octants, both end points drawn */
static void sd_line(uint32_t* img, int64_t Ht, int64_t Wt, int64_t i1, int64_t j1, int64_t i2, int64_t j2, uint32_t c)
{
    const int64_t di = llabs(i2 - i1), dj = -llabs(j2 - j1);
    const int64_t si = i1 < i2 ? 1 : -1, sj = j1 < j2 ? 1 : -1;
    int64_t err = di + dj;
    for (;;) {
        sd_put_pixel(img, Ht, Wt, i1, j1, c);
        if (i1 == i2 && j1 == j2) break;
        const int64_t e2 = 2 * err;
        if (e2 >= dj) { err += dj; i1 += si; }
        if (e2 <= di) { err += di; j1 += sj; }
    }
}
/* SD.Circle(position, diameter) with an odd diameter: ASSUMED midpoint circle of radius
 * diameter div 2 around position + radius, 8-way symmetric */
static void sd_circle(uint32_t* img, int64_t Ht, int64_t Wt, int64_t i_pos, int64_t j_pos, int64_t diameter, uint32_t c)
{
    const int64_t r = diameter / 2, ci = i_pos + r, cj = j_pos + r;
    int64_t x = 0, y = r, d = 1 - r;
    while (x <= y) {
        sd_put_pixel(img, Ht, Wt, ci + x, cj + y, c); sd_put_pixel(img, Ht, Wt, ci - x, cj + y, c);
        sd_put_pixel(img, Ht, Wt, ci + x, cj - y, c); sd_put_pixel(img, Ht, Wt, ci - x, cj - y, c);
        sd_put_pixel(img, Ht, Wt, ci + y, cj + x, c); sd_put_pixel(img, Ht, Wt, ci - y, cj + x, c);
        sd_put_pixel(img, Ht, Wt, ci + y, cj - x, c); sd_put_pixel(img, Ht, Wt, ci - y, cj - x, c);
        x += 1;
        if (d < 0) d += 2 * x + 1;
        else { y -= 1; d += 2 * (x - y) + 1; }
    }
}
/* wu_to_pu(x_wu, pu_per_wu) = floor(Int, x_wu * pu_per_wu) + 1  UT:6 (Float32 * Int -> Float32) */
static int64_t orc_wu_to_pu(real x, int32_t pu) { return (int64_t)R_FLOOR(x * (real)pu) + 1; }

/* update_top_view!(env)  SR:446-483 (+ draw_tile_map! SR:342-372) for agent a */
static void orc_update_top_view_agent(orc_batch* b, int32_t a)
{
    const int32_t H = b->H, W = b->W, N = b->N;
    const int64_t Ht = b->Ht, Wt = b->Wt;
    const size_t HW = (size_t)H * W;
    uint32_t* img = b->top_view + (size_t)Ht * Wt * a;
    const int64_t pu = Ht / H;                                          /* SR:346, SR:466 */
    const uint32_t colors[3] = {0x00FFFFFFu, 0x00FF0000u, 0x00000000u}; /* tile_map_colors SR:288 */
    for (int32_t j = 1; j <= W; ++j) {                                  /* SR:348 */
        for (int32_t i = 1; i <= H; ++i) {
            const int64_t i0 = (int64_t)(i - 1) * pu + 1, j0 = (int64_t)(j - 1) * pu + 1;   /* SR:350-351 */
            const size_t t = (size_t)(i - 1) + (size_t)H * (j - 1);
            /* findfirst over the objects: WALL, then GOAL, else colors[end]  SR:355-360 */
            const uint32_t c = b->wall[HW * a + t] ? colors[0] : (b->goalmap[HW * a + t] ? colors[1] : colors[2]);
            for (int64_t jj = j0; jj < j0 + pu; ++jj)                   /* SD.FilledRectangle SR:353,362 */
                for (int64_t ii = i0; ii < i0 + pu; ++ii) sd_put_pixel(img, Ht, Wt, ii, jj, c);
            for (int64_t jj = j0; jj < j0 + pu; ++jj) {                 /* SR:364-365 */
                sd_put_pixel(img, Ht, Wt, i0, jj, 0x00ccccccu);
                sd_put_pixel(img, Ht, Wt, i0 + pu - 1, jj, 0x00ccccccu);
            }
            for (int64_t ii = i0; ii < i0 + pu; ++ii) {                 /* SR:366-367 */
                sd_put_pixel(img, Ht, Wt, ii, j0, 0x00ccccccu);
                sd_put_pixel(img, Ht, Wt, ii, j0 + pu - 1, 0x00ccccccu);
            }
        }
    }
    const real px = b->pos[2 * a], py = b->pos[2 * a + 1];
    const int64_t ip = orc_wu_to_pu(px, (int32_t)pu), jp = orc_wu_to_pu(py, (int32_t)pu);     /* SR:468 */
    const int64_t rp = orc_wu_to_pu(CFG_RADIUS(&b->cfg), (int32_t)pu);                    /* SR:469 */
    for (int32_t i = 0; i < N; ++i) {                                   /* SR:473-477 */
        const size_t r = (size_t)N * a + i;
        const real ex = px + b->ray_dist[r] * b->ray_dirs[2 * r];
        const real ey = py + b->ray_dist[r] * b->ray_dirs[2 * r + 1];
        sd_line(img, Ht, Wt, ip, jp, orc_wu_to_pu(ex, (int32_t)pu), orc_wu_to_pu(ey, (int32_t)pu), 0x00808080u);
    }
    sd_circle(img, Ht, Wt, ip - rp, jp - rp, 2 * rp + 1, 0x00c0c0c0u);  /* SR:480 */
}

static void orc_render_agent(orc_batch* b, int32_t a)
{
    orc_cast_rays_agent(b, a);            /* SR:336 / SR:134 */
    if (b->top_view) orc_update_top_view_agent(b, a);   /* SR:337 / SR:328 */
    orc_update_camera_view_agent(b, a);   /* SR:338 / SR:329 */
}

/* world.reward::R (SR:33): zero(R) (SR:81,131,170-186) or goal_reward = one(R) (SR:82,167), in the batch's R */
static void orc_set_reward(orc_batch* b, int32_t a, int goal)
{
    switch (b->cfg.reward_type) {
    case RCW_REWARD_FLOAT64: ((double*)b->reward)[a] = goal ? b->cfg.goal_reward_f64 : 0.0; break;
    case RCW_REWARD_INT32:   ((int32_t*)b->reward)[a] = goal ? (int32_t)b->cfg.goal_reward_f64 : 0; break;
    case RCW_REWARD_INT64:   ((int64_t*)b->reward)[a] = goal ? (int64_t)b->cfg.goal_reward_f64 : 0; break;
    default:                 ((float*)b->reward)[a] = goal ? b->cfg.goal_reward : 0.0f; break;
    }
}

/* reset!(world)  SR:110-137 with the build's generator */
static void orc_reset_agent(orc_batch* b, int32_t a, uint64_t seed)
{
    const int32_t H = b->H, W = b->W;
    const size_t HW = (size_t)H * W;
    const uint64_t key = orc_episode_key(seed, (uint64_t)(b->cfg.agent_id_offset + a),
                                         (uint64_t)b->episode[a]);
    uint64_t n = 0;
    uint8_t* gm = b->goalmap + HW * a;
    const uint8_t* wm = b->wall + HW * a;
    /* tile_map[GOAL, goal_position] = false  SR:118 */
    gm[(b->goal[2 * a] - 1) + (size_t)H * (b->goal[2 * a + 1] - 1)] = 0;
    /* CartesianIndex(rand(2:H-1), rand(2:W-1))  SR:120 */
    const int32_t gi = 2 + (int32_t)orc_below(orc_draw(key, n++), (uint64_t)(H - 2));
    const int32_t gj = 2 + (int32_t)orc_below(orc_draw(key, n++), (uint64_t)(W - 2));
    b->goal[2 * a] = gi; b->goal[2 * a + 1] = gj;                    /* SR:121 */
    gm[(gi - 1) + (size_t)H * (gj - 1)] = 1;                         /* SR:122 */
    /* sample_empty_position(rng, tile_map)  UT:52-58 -> UT:23-37 */
    const uint64_t max_tries = (uint64_t)1024 * H * W;
    uint64_t lin = orc_below(orc_draw(key, n++), (uint64_t)HW);      /* UT:24 */
    int gave_up = 1;
    for (uint64_t t = 0; t < max_tries; ++t) {                       /* UT:26 */
        if (wm[lin] | gm[lin]) lin = orc_below(orc_draw(key, n++), (uint64_t)HW); /* UT:27-28 */
        else { gave_up = 0; break; }
    }
    /* UT:34 "@warn Could not sample an empty position in max_tries ... Returning non-empty position": the reference goes on */
    if (gave_up && b->status[a] == 0) b->status[a] = RCW_WARN_SAMPLER_GAVE_UP;
    const int32_t pi = (int32_t)(lin % (uint64_t)H) + 1, pj = (int32_t)(lin / (uint64_t)H) + 1;
    b->pos[2 * a] = (real)((double)pi - 0.5);                       /* SR:125 */
    b->pos[2 * a + 1] = (real)((double)pj - 0.5);
    b->dir[a] = (int32_t)orc_below(orc_draw(key, n++), (uint64_t)b->nd);   /* SR:128 */
    orc_set_reward(b, a, 0); b->done[a] = 0;                             /* SR:131-132 */
    b->episode[a] += 1;
    orc_render_agent(b, a);                                          /* SR:134, SR:329 */
}

ORC_EXPORT int orc_create(const rcw_config* cfg, int32_t batch, uint64_t seed, int render,
                          orc_batch** out)
{
    if (!cfg || !out || batch < 1) return RCW_ERR_INVALID_ARGUMENT;
    const int32_t H = cfg->height_tile_map_tu, W = cfg->width_tile_map_tu;
    if (H < 3 || W < 3 || cfg->num_rays < 1 || cfg->num_directions < 1 ||
        cfg->height_camera_view_pu < 1)
        return RCW_ERR_INVALID_ARGUMENT;
    orc_batch* b = (orc_batch*)calloc(1, sizeof *b);
    if (!b) return RCW_ERR_OUT_OF_MEMORY;
    b->cfg = *cfg; b->B = batch; b->H = H; b->W = W; b->N = cfg->num_rays;
    b->nd = cfg->num_directions; b->Hc = cfg->height_camera_view_pu; b->seed = seed;
    b->nchunks = (2 * H * W + 63) / 64; b->render = render;
    const size_t B = (size_t)batch, HW = (size_t)H * W, N = (size_t)b->N;
    b->directions = (real*)malloc(sizeof(real) * 2 * b->nd);
    b->ray_table = (real*)malloc(sizeof(real) * 2 * N * b->nd);
    b->wall = (uint8_t*)calloc(HW * B, 1);
    b->goalmap = (uint8_t*)calloc(HW * B, 1);
    b->pos = (real*)calloc(2 * B, sizeof(real));
    b->dir = (int32_t*)calloc(B, sizeof(int32_t));
    b->goal = (int32_t*)calloc(2 * B, sizeof(int32_t));
    b->reward = calloc(B, 8);
    b->done = (uint8_t*)calloc(B, 1);
    b->episode = (uint32_t*)calloc(B, sizeof(uint32_t));
    b->status = (int32_t*)calloc(B, sizeof(int32_t));
    b->ray_stop = (int64_t*)calloc(2 * N * B, sizeof(int64_t));
    b->ray_dim = (int64_t*)calloc(N * B, sizeof(int64_t));
    b->ray_dist = (real*)calloc(N * B, sizeof(real));
    b->ray_dirs = (real*)calloc(2 * N * B, sizeof(real));
    b->camera_view = (uint32_t*)calloc(render ? (size_t)b->Hc * N * B : 1, sizeof(uint32_t));
    b->Ht = H * cfg->pu_per_tu; b->Wt = W * cfg->pu_per_tu;
    if (cfg->render_top_view) {
        if (cfg->pu_per_tu < 1) { orc_destroy(b); return RCW_ERR_INVALID_ARGUMENT; }
        b->top_view = (uint32_t*)calloc((size_t)b->Ht * b->Wt * B, sizeof(uint32_t));
        if (!b->top_view) { orc_destroy(b); return RCW_ERR_OUT_OF_MEMORY; }
    }
    b->col_height = (int32_t*)calloc(N * B, sizeof(int32_t));
    b->col_colour = (uint8_t*)calloc(N * B, 1);
    if (!b->directions || !b->ray_table || !b->wall || !b->goalmap || !b->pos || !b->dir ||
        !b->goal || !b->reward || !b->done || !b->episode || !b->status || !b->ray_stop ||
        !b->ray_dim || !b->ray_dist || !b->ray_dirs || !b->camera_view || !b->col_height ||
        !b->col_colour) { orc_destroy(b); return RCW_ERR_OUT_OF_MEMORY; }
    orc_direction_table(b->nd, b->directions);
    orc_build_ray_table(b);
    for (size_t a = 0; a < B; ++a) {
        uint8_t* wm = b->wall + HW * a;
        /* wall ring SR:57-60 */
        for (int32_t i = 0; i < H; ++i) { wm[i] = 1; wm[i + (size_t)H * (W - 1)] = 1; }
        for (int32_t j = 0; j < W; ++j) { wm[(size_t)H * j] = 1; wm[(H - 1) + (size_t)H * j] = 1; }
        b->goal[2 * a] = 2; b->goal[2 * a + 1] = 2;   /* placeholder cleared by reset */
    }
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < batch; ++a) orc_reset_agent(b, a, seed);
    *out = b;
    return RCW_OK;
}

ORC_EXPORT int orc_set_direction_table(orc_batch* b, const real* dirs)
{
    memcpy(b->directions, dirs, sizeof(real) * 2 * b->nd);
    orc_build_ray_table(b);
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < b->B; ++a) orc_render_agent(b, a);
    return RCW_OK;
}

ORC_EXPORT int orc_reset(orc_batch* b, const uint8_t* mask, uint64_t seed)
{
    b->seed = seed;
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < b->B; ++a)
        if (!mask || mask[a]) orc_reset_agent(b, a, seed);
    return RCW_OK;
}

ORC_EXPORT int orc_set_state(orc_batch* b, const int32_t* goal_ij, const real* pos,
                             const int32_t* dir, const uint8_t* mask)
{
    const int32_t H = b->H, W = b->W;
    const size_t HW = (size_t)H * W;
    for (int32_t a = 0; a < b->B; ++a) {
        if (mask && !mask[a]) continue;
        const int32_t gi = goal_ij[2 * a], gj = goal_ij[2 * a + 1];
        if (gi < 2 || gi > H - 1 || gj < 2 || gj > W - 1) return RCW_ERR_INVALID_ARGUMENT;
        if (dir[a] < 0 || dir[a] >= b->nd) return RCW_ERR_INVALID_ARGUMENT;
        if (!isfinite(pos[2 * a]) || !isfinite(pos[2 * a + 1])) return RCW_ERR_INVALID_ARGUMENT;
        if (!(pos[2 * a] >= RC(1.0) && pos[2 * a] < (real)(H - 1) && pos[2 * a + 1] >= RC(1.0) &&
              pos[2 * a + 1] < (real)(W - 1))) return RCW_ERR_INVALID_ARGUMENT;
    }
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < b->B; ++a) {
        if (mask && !mask[a]) continue;
        uint8_t* gm = b->goalmap + HW * a;
        gm[(b->goal[2 * a] - 1) + (size_t)H * (b->goal[2 * a + 1] - 1)] = 0;   /* SR:118 */
        b->goal[2 * a] = goal_ij[2 * a]; b->goal[2 * a + 1] = goal_ij[2 * a + 1];
        gm[(b->goal[2 * a] - 1) + (size_t)H * (b->goal[2 * a + 1] - 1)] = 1;   /* SR:122 */
        b->pos[2 * a] = pos[2 * a]; b->pos[2 * a + 1] = pos[2 * a + 1];        /* SR:126 */
        b->dir[a] = dir[a];                                                     /* SR:129 */
        orc_set_reward(b, a, 0); b->done[a] = 0;                                    /* SR:131-132 */
        orc_render_agent(b, a);
    }
    return RCW_OK;
}

/* act!(world, action)  SR:139-191 for agent a */
static void orc_act_agent(orc_batch* b, int32_t a, int action)
{
    const int32_t H = b->H, W = b->W, nd = b->nd;
    const size_t HW = (size_t)H * W;
    if (action <= 2) {                                              /* SR:150 */
        const real d1 = b->directions[2 * b->dir[a]], d2 = b->directions[2 * b->dir[a] + 1];
        const real inc = CFG_INC(&b->cfg);
        real nx, ny;
        if (action == 1) { nx = b->pos[2 * a] + inc * d1; ny = b->pos[2 * a + 1] + inc * d2; } /* UT:16 */
        else             { nx = b->pos[2 * a] - inc * d1; ny = b->pos[2 * a + 1] - inc * d2; } /* UT:17 */
        const int g = orc_is_player_colliding(b->goalmap + HW * a, H, W, nx, ny,
                                              CFG_RADIUS(&b->cfg),
                                              b->cfg.out_of_bounds == RCW_OOB_TREAT_EMPTY);   /* SR:162 */
        const int w = orc_is_player_colliding(b->wall + HW * a, H, W, nx, ny,
                                              CFG_RADIUS(&b->cfg),
                                              b->cfg.out_of_bounds == RCW_OOB_TREAT_EMPTY);   /* SR:163 */
        if (g < 0 || w < 0) {   /* Julia: BoundsError before any mutation */
            b->status[a] = RCW_ERR_OUT_OF_BOUNDS;
            return;
        }
        if (g || w) {                                               /* SR:165 */
            if (g) { orc_set_reward(b, a, 1); b->done[a] = 1; }   /* SR:166-168 */
            else   { orc_set_reward(b, a, 0); b->done[a] = 0; }                 /* SR:170-171 */
        } else {
            b->pos[2 * a] = nx; b->pos[2 * a + 1] = ny;             /* SR:174 */
            orc_set_reward(b, a, 0); b->done[a] = 0;                    /* SR:175-176 */
        }
    } else {
        int32_t d = b->dir[a];
        if (action == 3) d = (d + 1) % nd;                          /* turn_left UT:13 */
        else             d = ((d - 1) % nd + nd) % nd;              /* turn_right UT:14 (floored mod) */
        b->dir[a] = d;                                              /* SR:185 */
        orc_set_reward(b, a, 0); b->done[a] = 0;                        /* SR:186-187 */
    }
}

/* act!(env, action)  SR:333-340 (minus update_top_view!) for the whole batch */
ORC_EXPORT int orc_step(orc_batch* b, const uint8_t* actions)
{
    for (int32_t a = 0; a < b->B; ++a)                              /* @assert SR:140 */
        if (actions[a] < 1 || actions[a] > RCW_NUM_ACTIONS) return RCW_ERR_INVALID_ACTION;
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < b->B; ++a) {
        if (b->cfg.auto_reset && b->done[a]) {
            orc_reset_agent(b, a, b->seed);
            continue;
        }
        orc_act_agent(b, a, actions[a]);    /* SR:335 */
        orc_render_agent(b, a);             /* SR:336, SR:338 */
    }
    return RCW_OK;
}

/* rcw_step_device semantics: an agent given an action outside 1..4 is not stepped (status
 * RCW_ERR_INVALID_ACTION), the others step. */
ORC_EXPORT int orc_step_lenient(orc_batch* b, const uint8_t* actions)
{
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < b->B; ++a) {
        if (actions[a] < 1 || actions[a] > RCW_NUM_ACTIONS) {
            b->status[a] = RCW_ERR_INVALID_ACTION;
            orc_render_agent(b, a);
            continue;
        }
        if (b->cfg.auto_reset && b->done[a]) { orc_reset_agent(b, a, b->seed); continue; }
        orc_act_agent(b, a, actions[a]);
        orc_render_agent(b, a);
    }
    return RCW_OK;
}

ORC_EXPORT void orc_clear_status(orc_batch* b) { memset(b->status, 0, sizeof(int32_t) * (size_t)b->B); }

/* ---- getters ----------------------------------------------------------------------- */
ORC_EXPORT const uint32_t* orc_camera_view(orc_batch* b) { return b->camera_view; }
ORC_EXPORT const uint32_t* orc_top_view(orc_batch* b) { return b->top_view; }
ORC_EXPORT const void* orc_reward(orc_batch* b) { return b->reward; }
ORC_EXPORT const uint8_t* orc_done(orc_batch* b) { return b->done; }
ORC_EXPORT const real* orc_position(orc_batch* b) { return b->pos; }
ORC_EXPORT const int32_t* orc_direction(orc_batch* b) { return b->dir; }
ORC_EXPORT const int32_t* orc_goal(orc_batch* b) { return b->goal; }
ORC_EXPORT const uint32_t* orc_episode(orc_batch* b) { return b->episode; }
ORC_EXPORT const int32_t* orc_status(orc_batch* b) { return b->status; }
ORC_EXPORT const int64_t* orc_ray_stop(orc_batch* b) { return b->ray_stop; }
ORC_EXPORT const int64_t* orc_ray_dim(orc_batch* b) { return b->ray_dim; }
ORC_EXPORT const real* orc_ray_dist(orc_batch* b) { return b->ray_dist; }
ORC_EXPORT const real* orc_ray_dirs(orc_batch* b) { return b->ray_dirs; }
ORC_EXPORT const int32_t* orc_col_height(orc_batch* b) { return b->col_height; }
ORC_EXPORT const uint8_t* orc_col_colour(orc_batch* b) { return b->col_colour; }
ORC_EXPORT const real* orc_directions(orc_batch* b) { return b->directions; }
ORC_EXPORT const real* orc_ray_table(orc_batch* b) { return b->ray_table; }
ORC_EXPORT int32_t orc_num_chunks(orc_batch* b) { return b->nchunks; }

/* tile_map as BitArray{3}(2, H, W).chunks per agent: UInt64 (nchunks, B)  SR:54 */
ORC_EXPORT void orc_tile_map_chunks(orc_batch* b, uint64_t* out)
{
    const int32_t H = b->H, W = b->W;
    const size_t HW = (size_t)H * W;
    memset(out, 0, sizeof(uint64_t) * (size_t)b->nchunks * b->B);
    for (int32_t a = 0; a < b->B; ++a)
        for (int32_t j = 0; j < W; ++j)
            for (int32_t i = 0; i < H; ++i) {
                const size_t t = (size_t)i + (size_t)H * j;
                const size_t bit = 2 * t;
                uint64_t* ch = out + (size_t)b->nchunks * a;
                if (b->wall[HW * a + t])    ch[bit >> 6] |= 1ULL << (bit & 63);
                if (b->goalmap[HW * a + t]) ch[(bit + 1) >> 6] |= 1ULL << ((bit + 1) & 63);
            }
}

ORC_EXPORT int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_EXPORT void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
