// update_top_view! SR:446-483, the two-kernel form's store kernels: the camera fill's moving window over the image with the top view's
// pixel logic (whole 256-row chunks, 8-pixel tiles, any pixel scale), with their launcher.  Overview: rcw_device.h.
#include "rcw_device.h"
#include "rcw_top.h"

#ifdef RCW_TRACE_WAVES
// Measurement build only (make trace, tools/wave_trace.py H,W,pu): rcw_top_store_flat_kernel's wavefronts leave what rcw_fill256_kernel's do
// (rcw_fill.hip) — in an array of this translation unit's own, with a reader of its own.
namespace { __device__ unsigned long long g_wave_trace[1024 * 40]; }
extern "C" __attribute__((visibility("default"))) int rcw_top_store_trace_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 1024 * 40);
}
#endif

namespace {

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

#ifdef RCW_DEV_SWITCHES
#include "dev/top_follow_wait.inc"   // RCW_TOP_FOLLOW, the store kernels wait for the draw kernel agent by agent (measured, rejected)
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

// ---- launcher -----------------------------------------------------------------------------------
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
