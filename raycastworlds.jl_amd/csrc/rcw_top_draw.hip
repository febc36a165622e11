// update_top_view! SR:446-483: the in-place kernel, the one-kernel ring, the draw kernel of the two-kernel form (alone and inside the
// camera fill's launch), with their launchers.  Overview: rcw_device.h.
#include "rcw_device.h"
#include "rcw_fill256.h"
#include "rcw_top.h"

namespace {

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
#include "dev/top_follow_publish.inc"   // RCW_TOP_FOLLOW, the draw kernel publishes its agents for a store kernel that follows it (measured, rejected)
#endif

#ifdef RCW_DEV_SWITCHES
#include "dev/top_draw_body_r4.inc"   // RCW_TOP_DRAW=r4, the round-4 body of the draw kernel
#endif

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

#ifdef RCW_DEV_SWITCHES
#include "dev/top_draw_step_no_lds.inc"   // RCW_TOP_DRAW=halfds / nods, the walk's step without its LDS atomic (timing probes)
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
#include "dev/top_draw_banks_start.inc"   // RCW_TOP_DRAW=banks, every lane starts its walk on its own LDS bank (measured, rejected)
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
#include "dev/top_draw_probe_loops.inc"   // RCW_TOP_DRAW=halfds / nods, the walk loops of the two timing probes
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

}  // namespace

// ---- launchers, and the geometry rules of the top view's forms ----------------------------------------------
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
           rcw_fill_takes_256(p, (long long)p.B * p.N) && (long long)p.fill_grid + p.B < (1ll << 31);
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
int rcw_top_draw_per_cu(const RcwDev& p, int draw_block, int lds_per_cu, int waves)
{
    const size_t lds = 4 * top_draw_lds_words(p);
    int n = lds ? (int)((size_t)lds_per_cu / lds) : 8;
    const int by_waves = waves / (draw_block / 64 > 0 ? draw_block / 64 : 1);
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
// Above 64 KiB of dynamic LDS a kernel has to be told so once (the CU has 160 KiB).  The attribute belongs to the
// FUNCTION, i.e. to every handle on the device: it is set once per device, to the fixed cap the geometry selection
// works with (16 B + 156 KiB for the ring kernel, 159 KiB for the draw kernel; the attribute itself is set to the CU's 160 KiB), never to one handle's own need — a
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
