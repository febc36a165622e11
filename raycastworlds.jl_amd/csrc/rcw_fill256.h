#pragma once
#include "rcw_device.h"

namespace {

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
#include "dev/fill256_extra_trips.inc"   // RCW_FILL_TRIPS: more dependent round trips in front of the prefetch
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

}  // namespace
