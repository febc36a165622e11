// Dev tool (not product): what HBM write rate does a pure fill of the observation batch
// reach on this chip, for the store shapes the step kernel could use?  Prints GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: one WG per 256 KiB frame, waves interleave 1 KiB columns (the step kernel's shape)
template <bool NT>
__global__ __launch_bounds__(256) void fill_wg_interleaved(u32x4* out, int vec_per_wg, uint32_t v)
{
    u32x4* base = out + (size_t)blockIdx.x * vec_per_wg;
    u32x4 val = {v, v + 1, v + 2, v + 3};
#pragma unroll 4
    for (int i = threadIdx.x; i < vec_per_wg; i += 256) {
        if (NT) __builtin_nontemporal_store(val, base + i); else base[i] = val;
    }
}
// B: one WG per frame, each wave owns a contiguous quarter
template <bool NT>
__global__ __launch_bounds__(256) void fill_wave_contig(u32x4* out, int vec_per_wg, uint32_t v)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per_wave = vec_per_wg / 4;
    u32x4* base = out + (size_t)blockIdx.x * vec_per_wg + (size_t)wave * per_wave;
    u32x4 val = {v, v + 1, v + 2, v + 3};
#pragma unroll 4
    for (int i = lane; i < per_wave; i += 64) {
        if (NT) __builtin_nontemporal_store(val, base + i); else base[i] = val;
    }
}
// C: classic grid-stride over the whole buffer
template <bool NT>
__global__ __launch_bounds__(256) void fill_grid_stride(u32x4* out, size_t nvec, uint32_t v)
{
    u32x4 val = {v, v + 1, v + 2, v + 3};
    const size_t stride = (size_t)gridDim.x * 256;
#pragma unroll 4
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        if (NT) __builtin_nontemporal_store(val, out + i); else out[i] = val;
    }
}
// D: persistent WGs, each loops over frames blockIdx, blockIdx+grid, ...
template <bool NT>
__global__ __launch_bounds__(256) void fill_persistent(u32x4* out, int vec_per_frame, int frames, uint32_t v)
{
    u32x4 val = {v, v + 1, v + 2, v + 3};
    for (int f = blockIdx.x; f < frames; f += gridDim.x) {
        u32x4* base = out + (size_t)f * vec_per_frame;
#pragma unroll 4
        for (int i = threadIdx.x; i < vec_per_frame; i += 256) {
            if (NT) __builtin_nontemporal_store(val, base + i); else base[i] = val;
        }
    }
}
// F: grid-stride with a configurable block (the rocclr memset shape: few WGs, moving window)
template <bool NT, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void fill_window(u32x4* out, size_t nvec, uint32_t v)
{
    u32x4 val = {v, v + 1, v + 2, v + 3};
    const size_t stride = (size_t)gridDim.x * BLOCK;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(val, out + i + u * stride); else out[i + u * stride] = val;
        }
    }
    for (; i < nvec; i += stride) { if (NT) __builtin_nontemporal_store(val, out + i); else out[i] = val; }
}
// G: the top view kernel's skeleton: persistent 512-thread WGs, threads 256..511 store frame after frame (wave w of
// the group writes the 1 KiB columns w, w+4, ...), threads 0..255 only meet them at one barrier per frame
template <bool NT, bool LDSREAD>
__global__ __launch_bounds__(512) void fill_top_skeleton(u32x4* out, int vec_per_frame, int frames, uint32_t v)
{
    __shared__ uint32_t plane[2304];
    const int role = threadIdx.x >> 8, tid = threadIdx.x & 255;
    if (LDSREAD) { for (int k = threadIdx.x; k < 2304; k += 512) plane[k] = v * k; __syncthreads(); }
    for (int f = blockIdx.x; f < frames; f += gridDim.x) {
        if (role == 1) {
            u32x4* base = out + (size_t)f * vec_per_frame;
            const int wave = tid >> 6, lane = tid & 63;
            for (int col = wave; col < vec_per_frame / 64; col += 4) {
                u32x4 val = {v, v + 1, v + 2, v + 3};
                if (LDSREAD) { const uint32_t w = plane[col * 9 + (lane >> 3)]; val.x ^= w; val.w += w >> 3; }
                u32x4* dst = base + (size_t)col * 64 + lane;
                if (NT) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
// Z: launch floor — a kernel that does nothing, at the cast kernel's grid shapes
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
// E: 1024-thread WG per frame
template <bool NT>
__global__ __launch_bounds__(1024) void fill_wg1024(u32x4* out, int vec_per_wg, uint32_t v)
{
    u32x4* base = out + (size_t)blockIdx.x * vec_per_wg;
    u32x4 val = {v, v + 1, v + 2, v + 3};
#pragma unroll 4
    for (int i = threadIdx.x; i < vec_per_wg; i += 1024) {
        if (NT) __builtin_nontemporal_store(val, base + i); else base[i] = val;
    }
}

template <typename F>
double time_it(F launch, hipStream_t s, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 4096;
    const int frame_bytes = argc > 2 ? atoi(argv[2]) : 262144;
    const size_t bytes = (size_t)frames * frame_bytes;
    const int vpf = frame_bytes / 16;
    const size_t nvec = bytes / 16;
    u32x4* buf; CK(hipMalloc(&buf, bytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int it = 50;
    auto rep = [&](const char* name, double ms) { printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6); fflush(stdout); };
    rep("hipMemsetD32Async", time_it([&](int i) { CK(hipMemsetD32Async((hipDeviceptr_t)buf, i, bytes / 4, s)); }, s, it));
    rep("A wg-interleaved nt", time_it([&](int i) { hipLaunchKernelGGL(fill_wg_interleaved<true>, dim3(frames), dim3(256), 0, s, buf, vpf, i); }, s, it));
    rep("A wg-interleaved plain", time_it([&](int i) { hipLaunchKernelGGL(fill_wg_interleaved<false>, dim3(frames), dim3(256), 0, s, buf, vpf, i); }, s, it));
    rep("B wave-contiguous nt", time_it([&](int i) { hipLaunchKernelGGL(fill_wave_contig<true>, dim3(frames), dim3(256), 0, s, buf, vpf, i); }, s, it));
    rep("B wave-contiguous plain", time_it([&](int i) { hipLaunchKernelGGL(fill_wave_contig<false>, dim3(frames), dim3(256), 0, s, buf, vpf, i); }, s, it));
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, sizeof nm, "C grid-stride nt grid=%d", g);
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL(fill_grid_stride<true>, dim3(g), dim3(256), 0, s, buf, nvec, i); }, s, it));
        snprintf(nm, sizeof nm, "C grid-stride plain grid=%d", g);
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL(fill_grid_stride<false>, dim3(g), dim3(256), 0, s, buf, nvec, i); }, s, it));
    }
    for (int g : {256, 512, 1024, 2048}) {
        char nm[64];
        snprintf(nm, sizeof nm, "D persistent nt grid=%d", g);
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL(fill_persistent<true>, dim3(g), dim3(256), 0, s, buf, vpf, frames, i); }, s, it));
    }
    for (int g : {768, 1024}) {
        char nm[64];
        snprintf(nm, sizeof nm, "G top skeleton nt grid=%d", g);
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL((fill_top_skeleton<true, false>), dim3(g), dim3(512), 0, s, buf, vpf, frames, i); }, s, it));
        snprintf(nm, sizeof nm, "G top skeleton nt+lds grid=%d", g);
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL((fill_top_skeleton<true, true>), dim3(g), dim3(512), 0, s, buf, vpf, frames, i); }, s, it));
        snprintf(nm, sizeof nm, "G top skeleton plain grid=%d", g);
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL((fill_top_skeleton<false, false>), dim3(g), dim3(512), 0, s, buf, vpf, frames, i); }, s, it));
    }
#define RUNF(NT, BLOCK, UNROLL, G) { char nm[80]; snprintf(nm, sizeof nm, "F window %s block=%d unroll=%d grid=%d", NT ? "nt" : "plain", BLOCK, UNROLL, G); \
        rep(nm, time_it([&](int i) { hipLaunchKernelGGL((fill_window<NT, BLOCK, UNROLL>), dim3(G), dim3(BLOCK), 0, s, buf, nvec, i); }, s, it)); }
    for (int g : {128, 256, 512, 1024}) { RUNF(false, 256, 1, g) RUNF(true, 256, 1, g) RUNF(false, 256, 4, g) RUNF(false, 1024, 1, g) RUNF(false, 1024, 4, g) RUNF(false, 512, 2, g) }
    for (int blk : {64, 128, 256}) { char nm[64]; snprintf(nm, sizeof nm, "Z empty kernel grid=4096 block=%d", blk);
        rep(nm, time_it([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(frames), dim3(blk), 0, s, (int*)nullptr); }, s, 200)); }
    rep("Z empty kernel grid=256 block=256", time_it([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (int*)nullptr); }, s, 200));
    rep("E wg1024 nt", time_it([&](int i) { hipLaunchKernelGGL(fill_wg1024<true>, dim3(frames), dim3(1024), 0, s, buf, vpf, i); }, s, it));
    rep("E wg1024 plain", time_it([&](int i) { hipLaunchKernelGGL(fill_wg1024<false>, dim3(frames), dim3(1024), 0, s, buf, vpf, i); }, s, it));
    CK(hipFree(buf));
    return 0;
}
