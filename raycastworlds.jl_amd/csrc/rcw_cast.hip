// The casting kernels of a step (rcw_cast_kernel; the one-launch step rcw_fill256_cast_kernel and what primes its slots) and the small
// state kernels (reset, set_state, tile map, ray materialisation), with their launchers.  Overview: rcw_device.h.
#include "rcw_device.h"
#include "rcw_fill256.h"

namespace {

#ifdef RCW_DEV_SWITCHES
#include "dev/cast_kernel_r3.inc"   // RCW_CAST_KERNEL=r3, the round-3 cast kernel and its two rejected variants
#endif

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
// The word the fill reads for a column: its padding (SR:436, 0..H_cam; H_cam <= 8191 here) | colour id << 13.
// column_padding(Hc, h) for 1 <= Hc <= 8191 in four 32-bit vector instructions (clamp, subtract, halve, clamp) instead of the general form's
// 64-bit difference, two compares and a divergent branch — five fans a column.  With g = clamp(h, -Hc - 2, Hc): Hc - g is in [0, 2 Hc + 2]; its
// half is 0 from h >= Hc - 1 on (SR:433), (Hc - h) / 2 in between (SR:436), and Hc + 1 -> Hc where the general form clamps
// (tests/test_host_logic.py replays both over the Int32 range).
__device__ __forceinline__ int spec_padding(int Hc, int h)
{
    const int g = min(max(h, -Hc - 2), Hc);
    return min((Hc - g) >> 1, Hc);
}
__device__ __forceinline__ uint32_t spec_word(int Hc, int h, int cid) { return (uint32_t)spec_padding(Hc, h) | ((uint32_t)cid << 13); }

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
        const uint16_t w = (uint16_t)spec_word(p.Hc, h, cid);
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
#include "dev/cast_waves_kernel.inc"   // RCW_CAST_WAVES=1, a wavefront per agent in the two-launch cast kernel (measured, rejected)
#endif

#ifdef RCW_DEV_SWITCHES
#include "dev/step256_kernel.inc"   // RCW_STEP_FUSED, cast and fill in one launch WITH a hand-off inside it (measured, rejected)
#endif

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
            pad_l = (int)(w & 0x1fffu);
            colour_l = p.colour[(w >> 13) & 3u];
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

// The same for the other camera heights the moving window takes (rcw_fill_window_kernel<M>, rcw_fill.hip): H_cam = 256 k (a chunk is one of
// the k row blocks of a column: M = 1), 128 or 64 (a chunk holds M = 2 or 4 whole columns, which start at a multiple of M: their M words are one
// 2 M-byte load).  The launcher takes this form below 2^31 chunks.
template <int M>
__device__ __forceinline__ void fill_window_spec_body(const RcwDev& p, const uint8_t* __restrict__ actions, const uint16_t* __restrict__ slots,
                                                      u32x4* __restrict__ out, long long total_chunks, int block, int blocks, int n_shift)
{
    const int lane = threadIdx.x & 63;
    const long long G = (long long)blocks * (kBlock / 64);
    const long long g = (long long)block * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ceil_c = p.ceiling_color, floor_c = p.floor_color;
    const int Hc = p.Hc;
    const uint32_t k = M == 1 ? (uint32_t)Hc >> 8 : 1u;   // chunks per column (M == 1)
    const int sub = M == 1 ? 0 : lane / (64 / M);          // this lane's column within a chunk (M > 1)
    const int r_lane = M == 1 ? lane * 4 : (lane - sub * (64 / M)) * 4;
    for (long long base = g; base < total_chunks; base += G * 64) {
        const long long mine = base + (long long)lane * G;
        int pad_l[M], rb_l = 0;
        uint32_t colour_l[M];
#pragma unroll
        for (int j = 0; j < M; ++j) { pad_l[j] = -1; colour_l[j] = 0u; }
        if (mine < total_chunks) {
            const uint32_t col0 = M == 1 ? (uint32_t)mine / k : (uint32_t)mine * (uint32_t)M;      // first (only) column of the chunk
            const uint32_t a = n_shift >= 0 ? col0 >> n_shift : col0 / (uint32_t)p.N;              // (a chunk never spans two agents: N H_cam % 256 == 0 here)
            rb_l = M == 1 ? (int)((uint32_t)mine - col0 * k) * 256 : 0;
            const uint32_t act = actions[a];
            asm volatile("" :: "v"(act) : "memory");
            const uint32_t sel = act - 1u < (uint32_t)RCW_NUM_ACTIONS ? act : 0u;
            struct __attribute__((aligned(2 * M))) Words { uint16_t w[M]; };
            const Words ws = *reinterpret_cast<const Words*>(slots + ((size_t)col0 + (size_t)(4u * a + sel) * (size_t)p.N));   // [B][5][N]
            asm volatile("" :: "v"(ws.w[0]) : "memory");
#pragma unroll
            for (int j = 0; j < M; ++j) {
                pad_l[j] = (int)(ws.w[j] & 0x1fffu);
                colour_l[j] = p.colour[(ws.w[j] >> 13) & 3u];
            }
        }
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
            const int pad0 = __builtin_amdgcn_readlane(pad_l[0], l);
            if (pad0 < 0) continue;            // wave-uniform: past the end
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

// The same launch for the other camera heights of the moving window (256 k, 128, 64 rows): the fill workgroups run rcw_fill_window_kernel<M>'s
// chunk logic (fill_window_spec_body<M>; M by a kernel argument: 1, 2 or 4).  A kernel of its own, so that rcw_fill256_cast_kernel — every
// BASELINE configuration's step — carries nothing but its own code.  (The 20-25 us its launch lost at 8192 agents x 1024 columns when these heights
// first went in was NOT these bodies, nor anything else in the source: the changed casting code had moved the fill's chunk loop off the start of a
// 64-byte line of the code — seven builds on one box, profiles/r06_step_forms.txt (5).  This file is compiled with -falign-loops=64 since.)
template <typename T, bool TIE_LE, bool DIST_PRE, bool WAVE>
__global__ __launch_bounds__(kBlock) void rcw_fill_window_cast_kernel(const RcwDev p, const uint8_t* __restrict__ actions, const uint8_t* __restrict__ mask,
                                                                      u32x4* __restrict__ out, long long total_chunks, int fill_blocks,
                                                                      const uint16_t* __restrict__ slots_in, uint16_t* __restrict__ slots_out, int lds_words, int n_shift, int cols,
                                                                      int window)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if ((int)blockIdx.x < fill_blocks) {
        if (window == 1) fill_window_spec_body<1>(p, actions, slots_in, out, total_chunks, (int)blockIdx.x, fill_blocks, n_shift);
        else if (window == 2) fill_window_spec_body<2>(p, actions, slots_in, out, total_chunks, (int)blockIdx.x, fill_blocks, n_shift);
        else fill_window_spec_body<4>(p, actions, slots_in, out, total_chunks, (int)blockIdx.x, fill_blocks, n_shift);
        return;
    }
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
#include "dev/launch_cast_waves.inc"   // RCW_CAST_WAVES=1, the launch of rcw_cast_waves_kernel
#endif
    RCW_DISPATCH(rcw_cast_kernel, dim3(count), dim3(p.cast_block), rcw_cast_lds_bytes(p), p, actions_dev, mask_dev, first);
    return hipGetLastError();
}

// The one-launch step (rcw_fill256_cast_kernel): the geometries that take it — a camera view the moving window of rcw_fill256_kernel /
// rcw_fill_window_kernel fills (256 k, 128 or 64 rows, up to 8191), no top view (its drawing needs the state the same launch commits), a
// batch of fewer than 2^29 view columns and 2^31 chunks — the bytes of ONE
// of its two slot buffers, and the launch: with_fill = the fill workgroups in front (a step); without, the casting workgroups alone
// (they prime the slots behind a reset / set_state, or for a first step: the camera fill then follows as a launch of its own).
int rcw_step_spec_eligible(const RcwDev& p)
{
    const long long cols = (long long)p.B * p.N;
    if (p.top_view != nullptr || p.fill_plain || cols >= (1ll << 29) || p.Hc > 8191) return 0;                  // (the slot word holds a padding of 13 bits)
    if (rcw_fill_window_columns(p, cols) < 0) return 0;                                                           // another fill kernel's camera height
    return cols * p.Hc / 256 < (1ll << 31) ? 1 : 0;                                                              // chunk ids in 32 bits
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
        const int window = rcw_fill_window_columns(p, total_cols);          // 0: the 256-row window; 1, 2, 4: rcw_fill_window_kernel<M>'s
        const long long units = window == 0 ? total_cols : total_cols * p.Hc / 256;   // view columns / 1 KiB chunks of the batch
        if (window < 0) return hipErrorInvalidValue;
        if (window == 0) {
            if (wave) RCW_DISPATCH_W(rcw_fill256_cast_kernel, true, dim3(fill_blocks + cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev,
                                     out, units, fill_blocks, slots_in, slots_out, lds_words, n_shift, icols);
            else      RCW_DISPATCH_W(rcw_fill256_cast_kernel, false, dim3(fill_blocks + cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev,
                                     out, units, fill_blocks, slots_in, slots_out, lds_words, n_shift, icols);
        } else {
            if (wave) RCW_DISPATCH_W(rcw_fill_window_cast_kernel, true, dim3(fill_blocks + cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev,
                                     out, units, fill_blocks, slots_in, slots_out, lds_words, n_shift, icols, window);
            else      RCW_DISPATCH_W(rcw_fill_window_cast_kernel, false, dim3(fill_blocks + cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev,
                                     out, units, fill_blocks, slots_in, slots_out, lds_words, n_shift, icols, window);
        }
    } else {
        if (wave) RCW_DISPATCH_W(rcw_cast_successors_kernel, true, dim3(cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev, slots_out, lds_words);
        else      RCW_DISPATCH_W(rcw_cast_successors_kernel, false, dim3(cast_blocks), dim3(kBlock), lds, p, actions_dev, mask_dev, slots_out, lds_words);
    }
    return hipGetLastError();
}

#ifdef RCW_DEV_SWITCHES
#include "dev/launch_step256.inc"   // RCW_STEP_FUSED, eligibility and launch of rcw_step256_kernel
#endif

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
