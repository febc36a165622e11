#pragma once
// What the top view's two translation units (rcw_top_draw.hip, rcw_top_store.hip) share.
#include "rcw_device.h"

// rcw_top_store_flat_kernel: the image columns a 256-pixel chunk can touch in this geometry; 0: the kernel does not take it
static size_t top_circle_table_bytes(const RcwDev& p) { return (size_t)(p.top_rp + 1) * ((2 * p.top_rp + 1 + 31) / 32 + 2) * 4; }
static size_t top_flat_plane_words(const RcwDev& p) { return (((size_t)p.H * p.pu * p.W * p.pu + 255 + 255) / 256) * 8; }
static size_t top_store_flat_lds_bytes(const RcwDev& p, int K)             // plane words | descriptors | circle rows | row table
{
    return (size_t)(kBlock / 64) * 512 * 4 + (size_t)(kBlock / 64) * 64 * K * 16 + ((top_circle_table_bytes(p) + 15) & ~(size_t)15) + (size_t)4 * p.H * p.pu;
}

namespace {

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

}  // namespace
