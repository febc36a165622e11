// The camera fill (update_camera_view! SR:431-440): the moving-window kernels and the two fallbacks, with their launchers.
// Overview: rcw_device.h.
#include "rcw_device.h"
#include "rcw_fill256.h"

namespace {

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
#include "dev/fill256_trips_kernel.inc"   // RCW_FILL_TRIPS, rcw_fill256_kernel's body with more round trips a prefetch
#endif

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

}  // namespace

// ---- launchers ----------------------------------------------------------------------------------
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
#include "dev/launch_fill256_trips.inc"   // RCW_FILL_TRIPS, the launch of rcw_fill256_trips_kernel
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

// (for the other translation units: does this geometry take rcw_fill256_kernel's window?)
int rcw_fill_takes_256(const RcwDev& p, long long total_cols) { return fill_choice(p, total_cols) == kFill256 ? 1 : 0; }
// ... or the window of rcw_fill_window_kernel<M>: 0 for the 256-row kernel, M = 1 / 2 / 4 for the window kernel's forms, -1 for the other fill kernels
int rcw_fill_window_columns(const RcwDev& p, long long total_cols)
{
    switch (fill_choice(p, total_cols)) {
    case kFill256: return 0;
    case kFillWindow1: return 1;
    case kFillWindow2: return 2;
    case kFillWindow4: return 4;
    default: return -1;
    }
}

hipError_t rcw_launch_expand(const RcwDev& p, const int32_t* col_h, const uint8_t* col_c,
                             int32_t count, uint32_t* frames, hipStream_t s)
{
    return rcw_launch_fill(p, col_h, col_c, frames, (long long)count * p.N, nullptr, s);
}
