// C ABI of librcw_hip (include/rcw.h): handle, HBM-resident state, host-built tables.
// Host code in this file that does floating point follows the reference operation for
// operation and must be compiled with -ffp-contract=off (see Makefile).
#include "../../include/rcw.h"
#include "rcw_kernels.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded at run time (load_rccl)

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#ifdef RCW_DEV_SWITCHES
// Development build only: which error returns of this file has the process taken?  Every `fail(...)` below leaves its source
// line in a table that rcw_dev_fail_sites hands out (tests: which refusals does the suite provoke, which never).
unsigned char g_fail_hit[4096];
template <typename... A>
int fail_at(int line, int code, const char* fmt, A... args)
{
    if (line >= 0 && line < (int)sizeof g_fail_hit) g_fail_hit[line] = 1;
    return fail(code, fmt, args...);
}
#define fail(...) fail_at(__LINE__, __VA_ARGS__)
#endif

#define RCW_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(e_ == hipErrorOutOfMemory ? RCW_ERR_OUT_OF_MEMORY : RCW_ERR_HIP, \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                 \
    } while (0)

}  // namespace
#ifdef RCW_DEV_SWITCHES
extern "C" __attribute__((visibility("default"))) int rcw_dev_fail_sites(unsigned char* out, int cap)
{
    if (!out || cap < 1) return -1;
    const int n = cap < (int)sizeof g_fail_hit ? cap : (int)sizeof g_fail_hit;
    std::memcpy(out, g_fail_hit, (size_t)n);
    return n;
}
#endif

struct rcw_handle {
    rcw_config cfg{};
    int32_t B = 0, device = 0, nchunks = 0, num_cus = 256;
    RcwHw hw{256, 160 * 1024, 32};     // the device's CUs, LDS bytes and wavefront slots a CU (hipDeviceProp_t: rcw_create)
    RcwDev dev{};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // device allocations
    void* d_pos = nullptr; void* d_dir = nullptr; void* d_goal = nullptr; void* d_reward = nullptr;
    void* d_done = nullptr; void* d_episode = nullptr; void* d_tile_map = nullptr;
    void* d_dir_table = nullptr; void* d_ray_table = nullptr; void* d_obs = nullptr;
    void* d_col_h = nullptr; void* d_col_c = nullptr; void* d_err = nullptr; void* d_status = nullptr;
    void* d_top_view = nullptr;
    // two-kernel top view: planes / player pixels / tile codes in HBM, the side stream the draw kernel runs on
    void* d_top_plane = nullptr; void* d_top_hdr = nullptr; void* d_top_codes = nullptr;
    void* d_top_flags = nullptr; uint32_t top_epoch = 0;   // the store kernel follows the draw kernel (dev/top_follow_publish.inc)
    // Several draw workgroups an agent (top_parts > 1) OR their bits into the agent's plane in HBM, and only rcw_top_store_kernel — which reads
    // every plane word exactly once — leaves the zero the next drawing needs: a drawing whose store did not follow (a failed launch in
    // between, the development build's skip-the-store switch) leaves bits behind that every later frame would carry.  Set in front of such a
    // drawing, cleared behind its store's launch; a drawing that finds it set clears the planes first.
    bool top_plane_dirty = false;
    hipStream_t top_stream = nullptr;
    hipEvent_t ev_top_fork = nullptr;
    hipEvent_t ev_top_join[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // one per run of agents
    void* d_actions = nullptr; void* d_mask = nullptr;
    void* d_in_goal = nullptr; void* d_in_pos = nullptr; void* d_in_dir = nullptr;
    int32_t* h_err = nullptr;   // pinned
    uint8_t* h_actions[2] = {nullptr, nullptr};   // pinned staging ring for rcw_step
    hipEvent_t ev_actions[2] = {nullptr, nullptr};
    int action_slot = 0;
    bool profiling = false;
    int step_pieces = 1;               // development experiment only (RCW_STEP_PIECES)
    void* d_step_flags = nullptr; void* d_step_hc = nullptr; uint32_t step_epoch = 0;   // development experiment only (RCW_STEP_FUSED)
    // the one-launch step (rcw_fill256_cast_kernel): two buffers of [B][5][N] packed column words — d_spec[spec_cur] holds the frames of the
    // CURRENT state (slot 0) and of its four successors (slots 1..4), written by the last casting launch; spec_primed: for every agent
    void* d_spec[2] = {nullptr, nullptr}; int spec_cur = 0; bool spec_primed = false;
    int spec_on = 0;                   // a step is ONE launch (rcw_fill256_cast_kernel)
    // The (height_line_pu, colour id) descriptors of the current frames (d_col_h / d_col_c) are what the two-launch step hands from its cast
    // kernel to its fill kernel; the one-launch step's fill reads the slots instead, and every store of the casting workgroups costs the
    // launch more than its bytes (profiles/r06_step_forms.txt) — so it writes the descriptors only for a caller that holds their device
    // pointers (cols_live: rcw_columns_device_ptr was called), and otherwise leaves them stale: ensure_columns recasts the current state
    // (the cast kernel, no action) in front of whatever reads them (rcw_columns, the gathers, rcw_update_camera_view).
    bool cols_live = false, cols_stale = false;
    int step_form_want = 0;            // rcw_set_step_form: 0 = the rule, or RCW_STEP_TWO_LAUNCHES / RCW_STEP_ONE_LAUNCH
    bool step_captured = false;        // a step of this handle was captured into a graph: it keeps the two-launch form from then on
    int prof_count = 0;
    std::vector<hipEvent_t> prof_ev;   // 4 per recorded step: start | after cast | after top view | after fill
    void* d_rays[4] = {nullptr, nullptr, nullptr, nullptr};   // rcw_rays scratch (grow-only)
    size_t rays_cap[4] = {0, 0, 0, 0};
    size_t reward_size = sizeof(float);
    // RCCL (loaded on demand): the observation gather
    void* comm = nullptr;              // ncclComm_t
    int32_t comm_rank = 0, comm_world = 0;
    void* d_gather_h = nullptr; void* d_gather_c = nullptr;   // gathered descriptors (B * world columns)
    bool real64 = false;            // world-unit type T = Float64 (cfg.world_unit_bits = 64)
    size_t real_size = sizeof(float);
    std::vector<float> dir_table;   // (2, nd)        T = Float32
    std::vector<float> ray_table;   // (N, 5, nd)
    std::vector<double> dir_table64;   //              T = Float64
    std::vector<double> ray_table64;
};

namespace {

constexpr int kProfileSlots = 256;

// update_top_view!(env) SR:446-483.  Two-kernel form: the draw kernel (VALU/LDS work, planes -> HBM) and the
// moving-window store kernel.  `between` (the camera fill, inside a step) is launched on the handle's stream while
// the draw kernel runs on the side stream: fork after what is already queued (the cast kernel), join before the store.
// The stand-alone call (`beside` = false) has no camera fill to run beside and takes the one-kernel form.
template <typename Between>
hipError_t launch_top_view_ordered(rcw_handle* h, const uint8_t* mask_dev, bool beside, Between between, hipEvent_t fused_event);
template <typename Between>
hipError_t launch_top_view(rcw_handle* h, const uint8_t* mask_dev, bool beside, Between between, hipEvent_t fused_event = nullptr)   // between(stream): the caller's camera fill
{
    const RcwDev& d = h->dev;
    hipError_t e;
    if (!d.top_split || (!beside && !d.top_alone_split)) {   // (nothing to hide the draw kernel behind: the one-kernel form is the faster one)
        if ((e = rcw_launch_top_view(d, mask_dev, h->stream)) != hipSuccess) return e;
        return between(h->stream);
    }
    if (d.top_parts > 1) {                                   // (see rcw_handle::top_plane_dirty; every order below forks from the handle's stream behind this)
        if (h->top_plane_dirty && (e = hipMemsetAsync(h->d_top_plane, 0, rcw_top_plane_bytes(d), h->stream)) != hipSuccess) return e;
        h->top_plane_dirty = true;
        struct Clean { rcw_handle* h; hipError_t* e; ~Clean() { if (*e == hipSuccess && !(h->dev.top_debug & 2)) h->top_plane_dirty = false; } };
        hipError_t result = hipErrorUnknown;
        Clean clean{h, &result};
        result = launch_top_view_ordered(h, mask_dev, beside, between, fused_event);
        return result;
    }
    return launch_top_view_ordered(h, mask_dev, beside, between, fused_event);
}

// (the launch orders of the two-kernel form; launch_top_view above decides whether it is taken)
template <typename Between>
hipError_t launch_top_view_ordered(rcw_handle* h, const uint8_t* mask_dev, bool beside, Between between, hipEvent_t fused_event)
{
    const RcwDev& d = h->dev;
    hipError_t e;
#ifdef RCW_DEV_SWITCHES
#include "dev/api_top_follow.inc"   // RCW_TOP_FOLLOW, the launch order in which the store kernel follows the draw kernel
#endif
    if (!beside) {                                           // stand-alone, two kernels back to back on the handle's stream
        if ((e = rcw_launch_top_draw(d, mask_dev, 0, d.B, h->stream, d.top_draw_block_alone)) != hipSuccess) return e;
        if ((e = rcw_launch_top_store(d, mask_dev, 0, d.B, h->stream)) != hipSuccess) return e;
        return between(h->stream);
    }
    if (d.top_fused) {
        // the camera fill and the drawing in ONE launch (rcw_fill256_draw_kernel), then the store: three launches on one
        // stream, no fork / join.  `between` — the camera fill of the caller — is replaced by that launch; its profiling
        // event (behind the fill, in front of the store kernel) is recorded here.
        if ((e = rcw_launch_fill256_draw(d, mask_dev, h->stream)) != hipSuccess) return e;
        if (fused_event && (e = hipEventRecord(fused_event, h->stream)) != hipSuccess) return e;
        return rcw_launch_top_store(d, mask_dev, 0, d.B, h->stream);
    }
    if (d.top_draw_first && d.top_runs <= 1) {
        // The DRAWING stays on the handle's stream, right behind the cast kernel, and the store kernel right behind the drawing; the camera
        // fill — which nothing of the top view depends on — goes to the side stream.  Measured with rocprofv3's kernel trace (tools/
        // step_timeline.sh): a kernel behind an event of the other stream starts ~13 us later than one behind a kernel of its own stream (19
        // against 6 us after the cast kernel's end), and the store kernel behind the join another 13 us after the drawing's end — with the
        // drawing on the side stream both lie on the step's critical path wherever the drawing outlasts the fill.  This way the late start
        // is the fill's, which has the drawing's whole time to spare, and the join at the end waits for a fill that ended long ago.
        if ((e = hipEventRecord(h->ev_top_fork, h->stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(h->top_stream, h->ev_top_fork, 0)) != hipSuccess) return e;
        e = between(h->top_stream);                              // (its profiling event is recorded on that stream too)
        const hipError_t rec = hipEventRecord(h->ev_top_join[0], h->top_stream);
        if (e == hipSuccess) e = rcw_launch_top_draw(d, mask_dev, 0, d.B, h->stream);
        if (e == hipSuccess) e = rcw_launch_top_store(d, mask_dev, 0, d.B, h->stream);
        if (rec == hipSuccess) { const hipError_t w = hipStreamWaitEvent(h->stream, h->ev_top_join[0], 0); if (e == hipSuccess) e = w; }
        return e == hipSuccess ? rec : e;
    }
    if ((e = hipEventRecord(h->ev_top_fork, h->stream)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(h->top_stream, h->ev_top_fork, 0)) != hipSuccess) return e;
    // The batch goes in d.top_runs runs of agents (one, unless the batch is several GiB of top view AND the drawing is
    // long against the camera fill): the side stream draws run after run without waiting for anything, the handle's
    // stream stores run r as soon as it is drawn — so what of the drawing does not fit beside the camera fill runs beside
    // the (HBM-bound) storing of earlier runs.
    // From here on the side stream may hold work: whatever fails, the handle's stream joins it again (every recorded
    // event is waited for), so that nothing runs on the side stream that the handle's stream does not wait for — a
    // later rcw_destroy / rcw_set_stream synchronises the handle's stream only, and a capture must end joined.
    const int runs = d.top_runs > 1 ? d.top_runs : 1;
    int recorded = 0;
    for (int r = 0; r < runs && e == hipSuccess; ++r) {
        const int first = (int)((long long)d.B * r / runs), count = (int)((long long)d.B * (r + 1) / runs) - first;
        e = rcw_launch_top_draw(d, mask_dev, first, count, h->top_stream);
        const hipError_t rec = hipEventRecord(h->ev_top_join[r], h->top_stream);    // (also after a failed launch: earlier runs' draws are queued)
        if (rec == hipSuccess) recorded = r + 1;
        if (e == hipSuccess) e = rec;
    }
    if (e == hipSuccess) e = between(h->stream);
    for (int r = 0; r < recorded; ++r) {
        const int first = (int)((long long)d.B * r / runs), count = (int)((long long)d.B * (r + 1) / runs) - first;
        const hipError_t w = hipStreamWaitEvent(h->stream, h->ev_top_join[r], 0);
        if (e == hipSuccess) e = w;
        if (e == hipSuccess) e = rcw_launch_top_store(d, mask_dev, first, count, h->stream);
    }
    return e;
}

// One step = cast kernel + fill kernel, back to back on the handle's stream (+ the top view when the handle renders
// it: before the fill with the one-kernel form, around it with the two-kernel form).  With profiling on,
// HIP events bracket each kernel (what bench.py's roofline block reads the fill kernel's
// duration from): start | after cast | after the top view (one-kernel form) or the fill (two-kernel form) | end.
hipError_t launch_step(rcw_handle* h, const uint8_t* actions_dev, const uint8_t* mask_dev)
{
    const RcwDev& d = h->dev;
    const bool prof = h->profiling && h->prof_count < kProfileSlots;
    hipEvent_t* ev = prof ? &h->prof_ev[4 * h->prof_count] : nullptr;
    hipError_t e;
    if (h->spec_on) {
        // The one-launch step keeps its place in the slot buffers on the HOST (which of the two the next launch reads): a graph would replay
        // one launch's pointers for ever.  A handle whose step is captured keeps the two-launch form from then on (replays advance the
        // state behind the library's back, so its slots can never be trusted again): rcw_step_form says so.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { h->spec_on = 0; h->step_captured = true; h->spec_primed = false; }
    }
    if (prof && (e = hipEventRecord(ev[0], h->stream)) != hipSuccess) return e;
    if (h->spec_on) {
        uint16_t* const cur = (uint16_t*)h->d_spec[h->spec_cur];
        if (actions_dev && !mask_dev && h->spec_primed) {
            // act!(env, a) SR:333-340 in ONE launch: the fill workgroups write the frames the actions select among the successors the last
            // casting launch left in `cur`; the casting workgroups commit the actions and cast the new states' successors into the other buffer
            uint16_t* const next = (uint16_t*)h->d_spec[h->spec_cur ^ 1];
            if (prof && ((e = hipEventRecord(ev[1], h->stream)) != hipSuccess || (e = hipEventRecord(ev[2], h->stream)) != hipSuccess)) return e;
            if ((e = rcw_launch_step_spec(d, actions_dev, nullptr, cur, next, true, h->cols_live, h->stream)) != hipSuccess) return e;
            h->spec_cur ^= 1;
            if (!h->cols_live) h->cols_stale = true;
            if (prof) { if ((e = hipEventRecord(ev[3], h->stream)) != hipSuccess) return e; h->prof_count++; }
            return hipSuccess;
        }
        // reset! / set_state (no action, maybe a mask) or a first step: the casting workgroups alone — dynamics if any, the current frame's
        // descriptors, and the (masked) agents' slots in place —, then the camera fill as a launch of its own
        if ((e = rcw_launch_step_spec(d, actions_dev, mask_dev, nullptr, cur, false, true, h->stream)) != hipSuccess) return e;
        if (!mask_dev) h->cols_stale = false;                 // (with a mask: the masked agents' descriptors are fresh — the fill below reads only those —, the others' as stale as before)
        if (!mask_dev) h->spec_primed = true;
        if (prof && ((e = hipEventRecord(ev[1], h->stream)) != hipSuccess || (e = hipEventRecord(ev[2], h->stream)) != hipSuccess)) return e;
        if ((e = rcw_launch_fill(d, d.col_h, d.col_c, d.obs, (long long)d.B * d.N, mask_dev, h->stream)) != hipSuccess) return e;
        if (prof) { if ((e = hipEventRecord(ev[3], h->stream)) != hipSuccess) return e; h->prof_count++; }
        return hipSuccess;
    }
#ifdef RCW_DEV_SWITCHES
#include "dev/api_step_pieces.inc"   // RCW_STEP_PIECES=2, the batch in two halves with the second cast beside the first fill
#endif
#ifdef RCW_DEV_SWITCHES
    if (d.step_fused && rcw_step_fusable(d)) {
        // Development experiment (RCW_STEP_FUSED=1, docs/experiments.md): cast and camera fill in ONE launch
        if (prof && ((e = hipEventRecord(ev[1], h->stream)) != hipSuccess || (e = hipEventRecord(ev[2], h->stream)) != hipSuccess)) return e;
        h->dev.step_epoch = ++h->step_epoch;
        if ((e = rcw_launch_step256(d, actions_dev, mask_dev, h->step_epoch, h->stream)) != hipSuccess) return e;
        if (prof) { if ((e = hipEventRecord(ev[3], h->stream)) != hipSuccess) return e; h->prof_count++; }
        return hipSuccess;
    }
#endif
    if ((e = rcw_launch_cast(d, actions_dev, mask_dev, h->stream)) != hipSuccess) return e;
    if (prof && (e = hipEventRecord(ev[1], h->stream)) != hipSuccess) return e;
    auto fill = [&](hipStream_t fs) -> hipError_t {            // (fs: the handle's stream, or its side stream: launch_top_view)
        hipError_t f;
        if (prof && !d.top_split && (f = hipEventRecord(ev[2], fs)) != hipSuccess) return f;
        if ((f = rcw_launch_fill(d, d.col_h, d.col_c, d.obs, (long long)d.B * d.N, mask_dev, fs)) != hipSuccess) return f;
        if (prof && d.top_split && (f = hipEventRecord(ev[2], fs)) != hipSuccess) return f;
        return hipSuccess;
    };
    if (d.top_view) { if ((e = launch_top_view(h, mask_dev, true, fill, prof ? ev[2] : nullptr)) != hipSuccess) return e; }   // SR:337
    else {
        if (prof && (e = hipEventRecord(ev[2], h->stream)) != hipSuccess) return e;
        if ((e = rcw_launch_fill(d, d.col_h, d.col_c, d.obs, (long long)d.B * d.N, mask_dev, h->stream)) != hipSuccess) return e;
    }
    if (prof) {
        if ((e = hipEventRecord(ev[3], h->stream)) != hipSuccess) return e;
        h->prof_count++;
    }
    return hipSuccess;
}

void free_all(rcw_handle* h)
{
    void** ptrs[] = {&h->d_pos, &h->d_dir, &h->d_goal, &h->d_reward, &h->d_done, &h->d_episode,
                     &h->d_tile_map, &h->d_dir_table, &h->d_ray_table, &h->d_obs, &h->d_col_h,
                     &h->d_col_c, &h->d_err, &h->d_status, &h->d_top_view, &h->d_actions, &h->d_mask, &h->d_in_goal,
                     &h->d_in_pos, &h->d_in_dir};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (h->h_err) (void)hipHostFree(h->h_err);
    h->h_err = nullptr;
    for (int k = 0; k < 2; ++k) {
        if (h->h_actions[k]) (void)hipHostFree(h->h_actions[k]);
        if (h->ev_actions[k]) (void)hipEventDestroy(h->ev_actions[k]);
        h->h_actions[k] = nullptr;
        h->ev_actions[k] = nullptr;
    }
    for (int k = 0; k < 4; ++k) { if (h->d_rays[k]) (void)hipFree(h->d_rays[k]); h->d_rays[k] = nullptr; h->rays_cap[k] = 0; }
    if (h->top_stream) (void)hipStreamSynchronize(h->top_stream);          // (a draw kernel of a failed step may still run)
    for (void** q : {&h->d_top_plane, &h->d_top_hdr, &h->d_top_codes, &h->d_top_flags, &h->d_step_flags, &h->d_step_hc, &h->d_spec[0], &h->d_spec[1]}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    if (h->ev_top_fork) (void)hipEventDestroy(h->ev_top_fork);
    for (hipEvent_t& q : h->ev_top_join) { if (q) (void)hipEventDestroy(q); q = nullptr; }
    if (h->top_stream) (void)hipStreamDestroy(h->top_stream);
    h->ev_top_fork = nullptr; h->top_stream = nullptr;
    if (h->d_gather_h) (void)hipFree(h->d_gather_h);
    if (h->d_gather_c) (void)hipFree(h->d_gather_c);
    h->d_gather_h = h->d_gather_c = nullptr;
    for (hipEvent_t ev : h->prof_ev) (void)hipEventDestroy(ev);
    h->prof_ev.clear();
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    h->ev_start = h->ev_stop = nullptr;
    h->own_stream = nullptr;
}

// directions_wu  SR:65-69: theta = (i-1)*2*pi/nd in Float64, components converted to T
template <typename T>
void build_direction_table(int nd, std::vector<T>& out)
{
    out.resize((size_t)2 * nd);
    for (int i = 1; i <= nd; ++i) {
        const double theta = (double)((long long)(i - 1) * 2) * 3.141592653589793 / (double)nd;
        out[2 * (size_t)(i - 1)] = (T)std::cos(theta);
        out[2 * (size_t)(i - 1) + 1] = (T)std::sin(theta);
    }
}

// Per heading d and ray i (SR:214-221, SR:404): the fan end points dir ± fov·rot₋₉₀(dir),
// the LinRange element (Float64 lerp converted to T), its normalisation, and the derived
// |1/dx|, |1/dy| (cast_ray's delta distances) and dir·ray (SR:404).
// Layout [nd][5][N]: see RCW_TABLE_ROWS.  T is the world-unit type; fov = convert(T, .) SR:267.
template <typename T>
void build_ray_table(const rcw_config& c, T fov, const std::vector<T>& dirs, std::vector<T>& out)
{
    const int N = c.num_rays, nd = c.num_directions;
    out.assign((size_t)nd * RCW_TABLE_ROWS * N, (T)0);
    const int lendiv = N - 1 > 1 ? N - 1 : 1;   // LinRange lendiv = max(len - 1, 1)
    for (int d = 0; d < nd; ++d) {
        const T d1 = dirs[2 * (size_t)d], d2 = dirs[2 * (size_t)d + 1];
        const T cam1 = d2, cam2 = -d1;                        // rotate_minus_90 SR:193
        const T fc1 = fov * cam1, fc2 = fov * cam2;
        const T first1 = d1 + fc1, first2 = d2 + fc2;         // SR:216
        const T last1 = d1 - fc1, last2 = d2 - fc2;           // SR:217
        T* row = out.data() + (size_t)d * RCW_TABLE_ROWS * N;
        for (int i = 0; i < N; ++i) {
            const double t = (double)i / (double)lendiv;      // lerpi: t = j/d in Float64
            const double omt = 1.0 - t;
            const double a1 = omt * (double)first1, b1 = t * (double)last1;
            const double a2 = omt * (double)first2, b2 = t * (double)last2;
            const T u1 = (T)(a1 + b1);
            const T u2 = (T)(a2 + b2);
            const T s1 = u1 * u1, s2 = u2 * u2;
            const T nrm = std::sqrt(s1 + s2);                 // norm(SVector) = sqrt(sum abs2)
            T r1, r2;
            if (c.normalize_mode == RCW_NORMALIZE_DIVIDE) {
                r1 = u1 / nrm; r2 = u2 / nrm;
            } else {
                const T inv = (T)1 / nrm;                     // inv(norm(a)) * a
                r1 = inv * u1; r2 = inv * u2;
            }
            const T m1 = d1 * r1, m2 = d2 * r2;               // sum(dir .* ray) SR:404
            row[i] = r1;
            row[(size_t)N + i] = r2;
            row[2 * (size_t)N + i] = std::fabs((T)1 / r1);
            row[3 * (size_t)N + i] = std::fabs((T)1 / r2);
            row[4 * (size_t)N + i] = m1 + m2;
        }
    }
}

void rebuild_ray_table(rcw_handle* h)
{
    if (h->real64) build_ray_table<double>(h->cfg, h->cfg.semi_field_of_view_wu_f64, h->dir_table64, h->ray_table64);
    else build_ray_table<float>(h->cfg, h->cfg.semi_field_of_view_wu, h->dir_table, h->ray_table);
}

int upload_tables(rcw_handle* h)
{
    const void* dirs = h->real64 ? (const void*)h->dir_table64.data() : (const void*)h->dir_table.data();
    const void* rays = h->real64 ? (const void*)h->ray_table64.data() : (const void*)h->ray_table.data();
    const size_t nd2 = (size_t)2 * h->cfg.num_directions, nr = (size_t)h->cfg.num_directions * RCW_TABLE_ROWS * h->cfg.num_rays;
    RCW_HIP(hipMemcpyAsync(h->d_dir_table, dirs, nd2 * h->real_size, hipMemcpyHostToDevice, h->stream));
    RCW_HIP(hipMemcpyAsync(h->d_ray_table, rays, nr * h->real_size, hipMemcpyHostToDevice, h->stream));
    RCW_HIP(hipStreamSynchronize(h->stream));
    return RCW_OK;
}

// Development switches: only a build with -DRCW_DEV_SWITCHES (make dev -> librcw_hip_dev.so) reads them.
#ifdef RCW_DEV_SWITCHES
#define RCW_DEV_ENV(name) std::getenv(name)
#else
#define RCW_DEV_ENV(name) (static_cast<const char*>(nullptr))
#endif

// The geometry of a batch as the kernels' argument block holds it: what the launchers' and the top view's rules read (rcw_create; the
// development build's rcw_dev_plan_top_view, which runs the rule without a device).
void set_geometry(RcwDev& d, const rcw_config* cfg, int32_t batch)
{
    d.B = batch; d.H = cfg->height_tile_map_tu; d.W = cfg->width_tile_map_tu; d.N = cfg->num_rays; d.nd = cfg->num_directions; d.Hc = cfg->height_camera_view_pu;
    d.real64 = cfg->world_unit_bits == 64 ? 1 : 0;
    d.pu = cfg->pu_per_tu;
    // player_radius_pu = wu_to_pu(player_radius_wu, pu_per_tu) SR:469 = floor(Int, r * pu) + 1 in T (UT:6)
    d.top_rp = d.real64 ? (int32_t)std::floor(cfg->player_radius_wu_f64 * (double)cfg->pu_per_tu) + 1
                        : (int32_t)std::floor(cfg->player_radius_wu * (float)cfg->pu_per_tu) + 1;
}

// ---- update_top_view! (SR:446-483): which form a handle takes — the RULES AS DATA ------------------------------------------------------
// Every threshold the choice of a form rests on, with the measurement that put it there.  The rule itself (top_view_rule below) is a pure
// function of the configuration, the batch and three numbers of the device (CUs, LDS and wavefronts a CU: rcw_create reads them from
// hipDeviceProp_t); tests/test_top_view_plan.py runs it on the CPU (development build: rcw_dev_plan_top_view) for every shape of the
// committed profile table and compares with tests/golden/top_view_plan_cases.json — the forms those profiles were taken with.  A retune
// on another box is an edit of this table, a re-run of tools/top_view_shapes.py and of tools/make_top_view_plan_cases.py; a change of a
// rule by accident is a red test.
struct TopRule { const char* name; double value; const char* unit; const char* evidence; };
enum TopRuleId {
    kRingThreeBuffersLds, kRingLdsCap, kRingWorkgroupsPerCu, kLineWalkMaxPixels, kAloneTwoKernelsPixels, kAloneTwoKernelsBelowPu,
    kDrawWideBlockLds, kDrawBlockMin, kDrawBlockMax, kAloneBlock64Agents, kAloneBlock128Agents, kRunsLineToCameraNum, kRunsLineToCameraDen,
    kRuns4Gib, kRuns2Gib, kSideStreamMinBytes, kPartsMax, kPartsMinRays, kFillGBperMs, kFillLateStartUs, kDrawUsPerGibFewRays,
    kDrawUsPerGibManyRays, kDrawManyRays, kDrawPartialRound, kDrawLdsCap, kFillWavefrontsPerCu, kTopRuleCount
};
constexpr TopRule kTopRules[kTopRuleCount] = {
    /* kRingThreeBuffersLds   */ {"ring_three_buffers_max_lds", 52 * 1024, "B", "profiles/r02_top_view_summary.txt: three workgroups of 8 wavefronts a CU still fit beside each other up to 52 KiB of ring each"},
    /* kRingLdsCap            */ {"ring_lds_cap", 156 * 1024, "B", "the CU's 160 KiB less what the kernel's static words and the runtime keep: beyond it the in-place form (profiles/r02_top_draw_lds.txt)"},
    /* kRingWorkgroupsPerCu   */ {"ring_workgroups_per_cu_max", 3, "", "profiles/r02_top_view_summary.txt: 3 x 8 wavefronts is what the ring kernel's register use admits; 4 measured no faster"},
    /* kLineWalkMaxPixels     */ {"line_walk_max_pixels", 16384, "px", "exactness, not tuning: the bit-plane kernels step a line on the carry of a 32-bit fraction, exact for lines of up to 2^14 pixels (tests/test_host_logic.py)"},
    /* kAloneTwoKernelsPixels */ {"stand_alone_two_kernels_from_pixels", 65536, "px", "profiles/r05_top_view_shapes.txt (b): draw -> store back to back 217 / 224 / 198 / 210 / 218 us/GiB against 214 / 231 / 228 / 253 / 360 for the ring from 256^2 px up"},
    /* kAloneTwoKernelsBelowPu*/ {"stand_alone_two_kernels_below_pu", 16, "px/tile", "profiles/r05_top_view_shapes.txt (b): 10 / 13 px a tile 360 / 302 against 507 / 450, 12 px 310 against 347; the ring keeps 16, 20, 24 ... px below 256^2 (264 / 228 / 229 against 268 / 246 / 233)"},
    /* kDrawWideBlockLds      */ {"draw_wide_block_from_plane_lds", 64 * 1024, "B", "profiles/r04_top_view_small_batches.txt, r03_top_view_shapes.txt: planes beyond 64 KiB leave one or two workgroups a CU: 512^2 px 180 / 182 / 200, 768^2 212 / 200 / 203, 1024^2 357 / 265 / 216 us with 256 / 512 / 1024 threads"},
    /* kDrawBlockMin          */ {"draw_wide_block_min_threads", 512, "threads", "same measurement"},
    /* kDrawBlockMax          */ {"draw_wide_block_max_threads", 1024, "threads", "same measurement (a lane per ray up to 1024 rays)"},
    /* kAloneBlock64Agents    */ {"stand_alone_64_threads_from_agents", 24576, "agents", "profiles/r05_draw_kernel.txt: 41,943 images of 80^2 px 175 us with 64 threads against 193 with 256"},
    /* kAloneBlock128Agents   */ {"stand_alone_128_threads_from_agents", 12288, "agents", "profiles/r05_draw_kernel.txt: 16,384 images of 128^2 px 103 us with 128 threads against 109"},
    /* kRunsLineToCameraNum   */ {"runs_when_lines_to_camera_num", 7, "", "profiles/r03_top_view_shapes.txt: (H + W) pu / 2 >= 1.75 H_cam, i.e. 2 (H + W) pu >= 7 H_cam: the drawing no longer fits beside the camera fill"},
    /* kRunsLineToCameraDen   */ {"runs_when_lines_to_camera_den", 2, "", "same rule's left-hand factor"},
    /* kRuns4Gib              */ {"four_runs_from_gib", 4, "GiB", "profiles/r03_top_view_shapes.txt: 16 GiB of top view 4516 / 4409 / 4332 / 4294 us with 1 / 2 / 4 / 8 runs, 32 GiB 8586 / 8459 / 7658 / 8068"},
    /* kRuns2Gib              */ {"two_runs_from_gib", 2, "GiB", "same table; runs of 256 MiB do not pay (205 vs 181 us at 1 GiB of 512^2 px images)"},
    /* kSideStreamMinBytes    */ {"side_stream_form_from_bytes", 256.0 * 1048576.0, "B", "profiles/r04_top_view_small_batches.txt: the fork / join and the extra launch cost ~13 us a step (39 / 51 / 53 / 60 / 102 / 341 us against the ring's 36 / 38 / 41 / 47 / 103 / 387 at 1 .. 4096 agents)"},
    /* kPartsMax              */ {"draw_parts_max", 4, "workgroups", "profiles/r05_draw_kernel.txt (tools/r05_draw_parts.sh): 1024^2 px x 64 agents 53.6 / 38.6 / 30.5 us with 1 / 2 / 4 parts"},
    /* kPartsMinRays          */ {"draw_part_min_rays", 128, "rays", "same table: a part's fixed costs (plane cleared, every end point, plane scanned) are most of a workgroup's life; x 256 agents 61.4 / 78.7 / 110"},
    /* kFillGBperMs           */ {"camera_fill_rate", 6.5e6, "B/us", "profiles/r05_kernel_stats.csv: rcw_fill256_kernel 156 us a GiB = 6.88 TB/s; 6.5 with its smaller siblings"},
    /* kFillLateStartUs       */ {"side_stream_late_start", 12, "us", "profiles/r05_top_view_shapes.txt / tools/step_timeline.sh: a kernel behind an event of the other stream starts ~13 us later than behind a kernel of its own (19 against 6 us after the cast kernel's end)"},
    /* kDrawUsPerGibFewRays   */ {"draw_floor_few_rays", 34, "us/GiB", "profiles/r05_draw_kernel.txt, r05_top_view_shapes.txt (a): the draw kernel's floor per GiB of top view with up to 256 rays (768^2 px x 455: 37 us)"},
    /* kDrawUsPerGibManyRays  */ {"draw_floor_many_rays", 55, "us/GiB", "same: beyond 256 rays (1024^2 px x 256, 1024 rays: 58-61 us)"},
    /* kDrawManyRays          */ {"draw_many_rays_from", 257, "rays", "the boundary between the two floors above"},
    /* kDrawPartialRound      */ {"draw_partial_round", 0.7, "", "profiles/r05_draw_kernel.txt (tools/r05_draw_first.sh): a partial round of draw workgroups takes about as long as a full one (768^2 px x 114 / 228 / 341 agents 90 -> 75, 129 -> 115, 169 -> 155 us)"},
    /* kDrawLdsCap            */ {"draw_kernel_lds_cap", 159 * 1024, "B", "rcw_top_split_unit / rcw_top_flat_cols: the draw kernel's plane + ray lists within the CU's LDS less 1 KiB"},
    /* kFillWavefrontsPerCu   */ {"fill_wavefronts_per_cu", 4, "wavefronts", "one workgroup of the camera fill (four wavefronts) sits on every CU: what is left of the CU's wavefront slots is the drawing's"},
};
constexpr double top_rule(TopRuleId id) { return kTopRules[id].value; }

// what the rule decides (fields of RcwDev), from the configuration, the batch, the device's numbers and the caller's wishes; no HIP call.
// want_form: 0 = the rule, or one of RCW_TOP_VIEW_IN_PLACE / ONE_KERNEL / TWO_KERNELS (rcw_set_top_view_form); want_runs: 0 = the rule, or 1..8.
// `lenient`: a form the geometry cannot take falls back to the rule (development switches) instead of failing.
int top_view_rule(RcwDev& d, const rcw_config* cfg, size_t B, const RcwHw& hw, int want_form, int want_runs, bool lenient)
{
    const int H = cfg->height_tile_map_tu, W = cfg->width_tile_map_tu, N = cfg->num_rays, Hc = cfg->height_camera_view_pu;
    d.top_blk_shift = 0; d.top_epoch = 0; d.top_signal = 0; d.top_follow = 0; d.top_follow_ok = 0;
    d.top_lds = 0; d.top_split = 0; d.top_flat = 0; d.top_plane_words = 0; d.top_unit_px = 256; d.top_runs = 1;
    d.top_alone_split = 0; d.top_fused = 0; d.top_grid = hw.cus; d.top_store_grid = d.fill_grid; d.top_store_plain = 0; d.top_draw_block = 256; d.top_draw_block_alone = 256; d.top_draw_first = 0; d.top_parts = 1;
    if (!cfg->render_top_view) {
        if (want_form != 0 && !lenient) return fail(RCW_ERR_UNSUPPORTED, "handle was created with render_top_view = 0");
        return RCW_OK;
    }
    const size_t ring_cap = (size_t)top_rule(kRingLdsCap);
    // the write-once kernel keeps a ring of 1..3 agents' bit planes in LDS: three where three workgroups per CU still fit beside
    // each other, else two, else one; larger images take the in-place kernel
    d.top_lds = 3;
    if (rcw_top_view_lds_bytes(d) > (size_t)top_rule(kRingThreeBuffersLds)) d.top_lds = 2;
    if (rcw_top_view_lds_bytes(d) > ring_cap) d.top_lds = 1;
    if (rcw_top_view_lds_bytes(d) > ring_cap) d.top_lds = 0;             // (the size depends on top_lds)
    if ((long long)H * cfg->pu_per_tu > (long long)top_rule(kLineWalkMaxPixels) || (long long)W * cfg->pu_per_tu > (long long)top_rule(kLineWalkMaxPixels)) d.top_lds = 0;
    if (const char* v = RCW_DEV_ENV("RCW_TOP_RING")) { const int k = std::atoi(v); if (k >= 1 && k <= 3 && d.top_lds > 0) { d.top_lds = k; if (rcw_top_view_lds_bytes(d) > ring_cap) d.top_lds = 1; } }
    if (want_form == RCW_TOP_VIEW_IN_PLACE) d.top_lds = 0;
    if (want_form == RCW_TOP_VIEW_ONE_KERNEL && !d.top_lds && !lenient)
        return fail(RCW_ERR_UNSUPPORTED, "the image's bit planes do not fit in LDS: this geometry takes the in-place form only");
    {   // persistent grid: as many 8-wavefront workgroups per CU as registers and LDS allow
        const size_t lds = rcw_top_view_lds_bytes(d);
        int per_cu = lds ? (int)((size_t)hw.lds_per_cu / lds) : 4;
        per_cu = per_cu < 1 ? 1 : (per_cu > (int)top_rule(kRingWorkgroupsPerCu) ? (int)top_rule(kRingWorkgroupsPerCu) : per_cu);
        d.top_grid = per_cu * hw.cus;
    }
    if (const char* v = RCW_DEV_ENV("RCW_TOP_GRID")) { const int g = std::atoi(v); if (g >= 1 && g <= 65536) d.top_grid = g; }
    if (const char* v = RCW_DEV_ENV("RCW_TOP_STORE_GRID")) { const int g = std::atoi(v); if (g >= 1 && g <= 65536) d.top_store_grid = g; }
    // The two-kernel form where the geometry allows it: the unit kernels (whole tiles in runs of 256 / 128 / 64 / 32 rows)
    // or the flat kernel (any pixel scale from 9, any image height that is a multiple of 4 from 42 rows)
    int unit = d.top_lds > 0 ? rcw_top_split_unit(d) : 0;
    const int flat = d.top_lds > 0 ? rcw_top_flat_cols(d) : 0;
    // (several units a chunk: the flat kernel is the faster one — 384² / 320² / 288² px images, µs per GiB: 181 / 194 / 207 with
    // 2 / 4 / 8 units against 176 / 175 / 173; whole 256-row chunks keep rcw_top_store_kernel: 159 against 179)
    if (flat && unit && unit < 256) unit = 0;
    if (const char* v = RCW_DEV_ENV("RCW_TOP_FLAT")) { const int f = std::atoi(v); if (f == 1 && flat) unit = 0; if (f == 0 && rcw_top_split_unit(d) && d.top_lds > 0) unit = rcw_top_split_unit(d); }
    const bool eligible = unit || flat;
    d.top_unit_px = unit ? unit : 256;
    d.top_flat = unit ? 0 : flat;
    d.top_plane_words = d.top_flat ? rcw_top_plane_words(d) : 0;
    // ... at every batch size where a step's camera fill and the drawing go in ONE launch (rcw_fill256_draw_kernel); where the drawing
    // needs the side stream (another camera height, planes beyond 64 KiB, runs of agents), only where the batch is big enough to pay for
    // the fork / join and the extra launch (kSideStreamMinBytes).  (Decided below, when the draw kernel's block and the runs are known.)
    d.top_split = eligible ? 1 : 0;
    if (want_form == RCW_TOP_VIEW_ONE_KERNEL || want_form == RCW_TOP_VIEW_IN_PLACE) d.top_split = 0;
    if (want_form == RCW_TOP_VIEW_TWO_KERNELS) {
        if (eligible) d.top_split = 1;
        else if (!lenient) return fail(RCW_ERR_UNSUPPORTED, "this geometry does not take the two-kernel form (pu_per_tu >= 8, image height a multiple of 4 and of at least 42 rows, bit plane within LDS)");
    }
    if (!d.top_split) { d.top_unit_px = 256; d.top_flat = 0; d.top_plane_words = 0; }
    // rcw_update_top_view alone has no camera fill to hide the drawing behind (kAloneTwoKernelsPixels, kAloneTwoKernelsBelowPu): draw ->
    // store back to back for images from 256 x 256 px, pixel scales that are no multiple of 4 and tiles below 16 px; the one-kernel form
    // keeps what is left of the two-kernel form's geometries — and every geometry the two-kernel form cannot take.
    {
        const long long px = (long long)H * cfg->pu_per_tu * W * cfg->pu_per_tu;
        d.top_alone_split = d.top_split && (px >= (long long)top_rule(kAloneTwoKernelsPixels) || (cfg->pu_per_tu & 3) != 0 || cfg->pu_per_tu < (int)top_rule(kAloneTwoKernelsBelowPu)) ? 1 : 0;
    }
    if (const char* v = RCW_DEV_ENV("RCW_TOP_ALONE_SPLIT")) d.top_alone_split = d.top_split && std::atoi(v) ? 1 : 0;
    // draw kernel: one workgroup of 4 wavefronts per agent; where the bit plane leaves room for one or two workgroups on a CU
    // (kDrawWideBlockLds), 8 to 16 wavefronts: a lane per ray for N > 256, two lanes a ray for fewer
    if (rcw_top_view_lds_bytes(d) / (d.top_lds > 0 ? d.top_lds : 1) > (size_t)top_rule(kDrawWideBlockLds)) {
        const int b = ((N + 255) / 256) * 256;
        d.top_draw_block = b < (int)top_rule(kDrawBlockMin) ? (int)top_rule(kDrawBlockMin) : (b > (int)top_rule(kDrawBlockMax) ? (int)top_rule(kDrawBlockMax) : b);
    }
    // ... and alone, with tens of thousands of small images, one or two wavefronts an agent (the set-up per wavefront is what such a batch costs)
    d.top_draw_block_alone = d.top_draw_block;
    if (d.top_draw_block == 256) d.top_draw_block_alone = B >= (size_t)top_rule(kAloneBlock64Agents) ? 64 : (B >= (size_t)top_rule(kAloneBlock128Agents) ? 128 : 256);
    if (const char* v = RCW_DEV_ENV("RCW_TOP_DRAW_BLOCK")) { const int b = std::atoi(v); if (b == 64 || b == 128 || b == 256 || b == 512 || b == 768 || b == 1024) d.top_draw_block = d.top_draw_block_alone = b; }
    // runs of agents: where the lines are long against the camera image's columns the drawing does not fit beside the camera fill; with
    // several GiB of top view a step, runs of >= 1 GiB let the rest of it hide beside the storing of earlier runs
    if ((long long)top_rule(kRunsLineToCameraDen) * ((long long)H + W) * cfg->pu_per_tu >= (long long)top_rule(kRunsLineToCameraNum) * Hc) {
        const size_t gib = (B * (size_t)H * W * cfg->pu_per_tu * cfg->pu_per_tu * sizeof(uint32_t)) >> 30;
        d.top_runs = gib >= (size_t)top_rule(kRuns4Gib) ? 4 : (gib >= (size_t)top_rule(kRuns2Gib) ? 2 : 1);
    }
    if (want_runs >= 1) d.top_runs = want_runs <= 8 ? (want_runs <= (int)B ? want_runs : (int)B) : 8;
    // a step's camera fill and the drawing in one launch where the geometry allows (256-row camera view, one run, planes of a
    // 256-thread draw workgroup): no side stream in the step
    d.top_fused = rcw_fill_draw_fusable(d) ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_TOP_FUSED")) d.top_fused = d.top_fused && std::atoi(v) ? 1 : 0;
    if (d.top_split && !d.top_fused && want_form != RCW_TOP_VIEW_TWO_KERNELS &&
        (double)(B * (size_t)H * W * cfg->pu_per_tu * cfg->pu_per_tu * sizeof(uint32_t)) < top_rule(kSideStreamMinBytes)) {
        d.top_split = 0; d.top_unit_px = 256; d.top_flat = 0; d.top_plane_words = 0; d.top_alone_split = 0;
    }
    // Several draw workgroups an agent (rcw_top_draw_kernel: each walks a part of the fan and ORs its plane into the agent's) where a batch
    // of big images leaves draw slots empty: as many parts as fill them (kPartsMax, kPartsMinRays).  Only with rcw_top_store_kernel, which
    // reads every plane word exactly once and leaves the zero the next drawing needs.
    const int draw_per_cu = rcw_top_draw_per_cu(d, d.top_draw_block, hw.lds_per_cu, hw.waves_per_cu - (int)top_rule(kFillWavefrontsPerCu));
    d.top_parts = 1;
    if (d.top_split && !d.top_flat && d.top_unit_px == 256 && !d.top_fused && !d.top_draw_r4) {
        const long long slots = (long long)hw.cus * draw_per_cu;
        int parts = (int)std::min<long long>((long long)top_rule(kPartsMax), slots / (long long)B);
        while (parts > 1 && N / parts < (int)top_rule(kPartsMinRays)) --parts;
        d.top_parts = parts < 1 ? 1 : parts;
        if (const char* v = RCW_DEV_ENV("RCW_TOP_PARTS")) { const int q = std::atoi(v); if (q >= 1 && q <= 4 && N / q >= 16) d.top_parts = q; }
    }
    // The drawing first on the handle's stream and the camera fill on the side stream (launch_top_view) where the fill is the SHORTER of the
    // two: it then ends before the store kernel starts (where it is the longer one it runs into the store kernel — two moving windows on one
    // HBM — and the step takes up to 60 % longer).  Both are estimated from the sizes: the fill at kFillGBperMs plus its late start, the
    // drawing at its measured floor per GiB of top view — of the batch or, for a small one, of most of one round of workgroups.
    {
        const double fill_us = (double)B * N * Hc * 4.0 / top_rule(kFillGBperMs) + top_rule(kFillLateStartUs);
        const double image_gib = (double)H * W * cfg->pu_per_tu * cfg->pu_per_tu * 4.0 / (double)(1u << 30);
        const double round_gib = (double)hw.cus * draw_per_cu * image_gib;
        const double top_gib = std::max((double)B * image_gib, top_rule(kDrawPartialRound) * round_gib);
        const double draw_us = top_gib * (N >= (int)top_rule(kDrawManyRays) ? top_rule(kDrawUsPerGibManyRays) : top_rule(kDrawUsPerGibFewRays));
        d.top_draw_first = d.top_split && !d.top_fused && d.top_runs <= 1 && fill_us <= draw_us ? 1 : 0;
    }
    if (const char* v = RCW_DEV_ENV("RCW_TOP_DRAW_FIRST")) d.top_draw_first = d.top_split && !d.top_fused && std::atoi(v) ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_TOP_STORE_PLAIN")) d.top_store_plain = std::atoi(v) ? 1 : 0;
    return RCW_OK;
}

// Which form update_top_view! (SR:446-483) takes for this handle (top_view_rule), and its scratch in HBM.
int plan_top_view(rcw_handle* h, int want_form, int want_runs, bool lenient)
{
    RcwDev& d = h->dev;
    const size_t B = (size_t)h->B;
    if (h->top_stream) RCW_HIP(hipStreamSynchronize(h->top_stream));
    for (void** q : {&h->d_top_plane, &h->d_top_hdr, &h->d_top_codes, &h->d_top_flags}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    d.top_plane = nullptr; d.top_hdr = nullptr; d.top_codes = nullptr; d.top_flags = nullptr; h->top_epoch = 0;
    int rc = top_view_rule(d, &h->cfg, B, h->hw, want_form, want_runs, lenient);
    if (rc != RCW_OK || !h->cfg.render_top_view) return rc;
    if (d.top_split) {
        hipError_t e = hipMalloc(&h->d_top_plane, rcw_top_plane_bytes(d));
        // The planes start out ZERO.  The flat store kernel ORs the plane words of two neighbouring agents' regions in a chunk
        // that holds pixels of both and relies on a region's bits outside its own image being zero — true of every region the
        // draw kernel has written, but a masked render right after rcw_set_top_view_form (whose own re-render may be the
        // one-kernel form, which writes no planes) draws the masked agents only and reads their neighbours' regions as they lie.
        // (stream-ordered on the handle's stream: every later launch of the handle comes behind it, the side stream's draw
        // kernel through the fork event)
        if (e == hipSuccess) e = hipMemsetAsync(h->d_top_plane, 0, rcw_top_plane_bytes(d), h->stream);
        if (e == hipSuccess) e = hipMalloc(&h->d_top_hdr, (size_t)h->B * sizeof(int2));
        if (e == hipSuccess) e = hipMemsetAsync(h->d_top_hdr, 0, (size_t)h->B * sizeof(int2), h->stream);
        if (e == hipSuccess) e = hipMalloc(&h->d_top_codes, rcw_top_codes_bytes(d));
#ifdef RCW_DEV_SWITCHES
        if (e == hipSuccess) e = hipMalloc(&h->d_top_flags, (size_t)h->B * sizeof(uint32_t));                      // (experiment RCW_TOP_FOLLOW)
        if (e == hipSuccess) e = hipMemsetAsync(h->d_top_flags, 0, (size_t)h->B * sizeof(uint32_t), h->stream);
#endif
        if (e == hipSuccess && !h->top_stream) e = hipStreamCreateWithFlags(&h->top_stream, hipStreamNonBlocking);
        if (e == hipSuccess && !h->ev_top_fork) e = hipEventCreateWithFlags(&h->ev_top_fork, hipEventDisableTiming);
        for (hipEvent_t& q : h->ev_top_join) if (e == hipSuccess && !q) e = hipEventCreateWithFlags(&q, hipEventDisableTiming);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RCW_ERR_OUT_OF_MEMORY : RCW_ERR_HIP, "top view planes: %s", hipGetErrorString(e));
        d.top_plane = (uint32_t*)h->d_top_plane; d.top_hdr = (int2*)h->d_top_hdr; d.top_codes = (uint2*)h->d_top_codes;
#ifdef RCW_DEV_SWITCHES
#include "dev/api_top_follow_plan.inc"   // RCW_TOP_FOLLOW: the counters' block size and where the store kernel may follow the draw kernel
#endif
    }
    hipError_t e = rcw_prepare_top_view(d, h->device);
    if (e != hipSuccess) return fail(RCW_ERR_HIP, "top view kernel attribute: %s", hipGetErrorString(e));
    return RCW_OK;
}

// The one-launch step pays where the fill outlasts the casting half's own life: one casting workgroup marches FIVE fans one after the
// other, so a small batch waits for it (4096 x 256 columns: 17 us of casting life under a 154 us fill; 64 agents: 22 us a step against
// 12 for cast kernel + fill).  Measured crossovers (profiles/r06_small_batches.txt, frames of a step): 8x8 map, 256 columns ~128 MiB;
// 16x16, 512 columns ~100 MiB; 32x32, 1024 columns ~350 MiB; 8x8, 64 columns below 64 MiB — the casting life fits
// kStepCastBaseUs + kStepCastUsPerUnit x (view columns a lane x 5 fans x (H + W) tiles a ray may cross), the fill kFillGBperMs.
constexpr double kStepCastBaseUs = 4.5, kStepCastUsPerUnit = 0.045;
bool step_one_launch_pays(const RcwDev& d)
{
    const int lanes = d.N <= 256 ? 64 : 256;                                // a wavefront per agent up to 256 view columns, a workgroup beyond
    const double units = (double)((d.N + lanes - 1) / lanes) * 5.0 * (double)(d.H + d.W);
    const double cast_us = kStepCastBaseUs + kStepCastUsPerUnit * units;
    const double fill_us = (double)d.B * d.N * d.Hc * 4.0 / top_rule(kFillGBperMs);
    return fill_us >= cast_us;
}

// Which form a step takes (rcw_set_step_form; want = 0: the rule — one launch where the geometry allows AND the batch is large enough
// for it to pay, unless a step of the handle was captured into a graph).  Allocates the two slot buffers the first time the one-launch
// form is taken; the caller primes them (launch_step without an action).
int plan_step_form(rcw_handle* h, int want)
{
    RcwDev& d = h->dev;
    const bool eligible = rcw_step_spec_eligible(d) != 0;
    if (want == RCW_STEP_ONE_LAUNCH && !eligible)
        return fail(RCW_ERR_UNSUPPORTED, "this handle does not take the one-launch step (a camera view of 256 k, 128 or 64 rows — up to 8191 — without a top view, fewer than 2^29 view columns)");
    const bool on = want == RCW_STEP_TWO_LAUNCHES ? false : (want == RCW_STEP_ONE_LAUNCH ? true : eligible && !h->step_captured && step_one_launch_pays(d));
    if (on) {
        for (int k = 0; k < 2; ++k) {
            if (h->d_spec[k]) continue;
            const hipError_t e = hipMalloc(&h->d_spec[k], rcw_step_spec_slot_bytes(d));
            if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RCW_ERR_OUT_OF_MEMORY : RCW_ERR_HIP, "one-launch step, slot buffers: %s", hipGetErrorString(e));
        }
        if (want == RCW_STEP_ONE_LAUNCH) h->step_captured = false;
    }
    if (on && !h->spec_on) h->spec_primed = false;
    h->spec_on = on ? 1 : 0;
    h->step_form_want = want;
    return RCW_OK;
}

#ifdef RCW_DEV_SWITCHES
#include "dev/api_plan_export.inc"   // rcw_dev_plan_top_view / rcw_dev_top_view_rules / rcw_dev_step_rule: the rules without a device (tests/test_top_view_plan.py)
#endif

int validate_config(const rcw_config* c, int32_t batch)
{
    if (c->abi_version != RCW_ABI_VERSION)
        return fail(RCW_ERR_INVALID_ARGUMENT, "rcw_config.abi_version %d != %d", c->abi_version, RCW_ABI_VERSION);
    if (batch < 1) return fail(RCW_ERR_INVALID_ARGUMENT, "batch must be >= 1 (got %d)", batch);
    if (c->height_tile_map_tu < 3 || c->width_tile_map_tu < 3)
        return fail(RCW_ERR_INVALID_ARGUMENT, "tile map must be at least 3x3 (got %dx%d)",
                    c->height_tile_map_tu, c->width_tile_map_tu);
    // the cast kernel stages a byte per tile in dynamic LDS next to a few static words: 64 KiB per workgroup in all
    if ((long long)c->height_tile_map_tu * c->width_tile_map_tu + 2ll * c->height_tile_map_tu > 65536 - 256)   // (+ the cast kernel's two guard bands of H bytes)
        return fail(RCW_ERR_UNSUPPORTED, "tile map larger than 65280 tiles does not fit the LDS staging");
    if (c->num_directions < 1 || c->num_rays < 1 || c->height_camera_view_pu < 1)
        return fail(RCW_ERR_INVALID_ARGUMENT, "num_directions, num_rays, height_camera_view_pu must be >= 1");
    if (c->num_rays > (1 << 24)) return fail(RCW_ERR_INVALID_ARGUMENT, "num_rays not exactly representable in Float32");
    if (c->num_directions > (1 << 20) || (long long)c->num_directions * c->num_rays > (1ll << 24))
        return fail(RCW_ERR_UNSUPPORTED, "num_directions * num_rays = %lld: the (direction, ray) table is limited to 2^24 entries",
                    (long long)c->num_directions * c->num_rays);
    if (c->height_camera_view_pu > (1 << 20))
        return fail(RCW_ERR_UNSUPPORTED, "height_camera_view_pu larger than 2^20");
    if (c->reward_type < RCW_REWARD_FLOAT32 || c->reward_type > RCW_REWARD_INT64)
        return fail(RCW_ERR_INVALID_ARGUMENT, "reward_type must be one of RCW_REWARD_* (got %d)", c->reward_type);
    if (!std::isfinite(c->goal_reward) || !std::isfinite(c->goal_reward_f64))
        return fail(RCW_ERR_INVALID_ARGUMENT, "goal_reward must be finite");
    if ((c->reward_type == RCW_REWARD_INT32 || c->reward_type == RCW_REWARD_INT64) &&
        (c->goal_reward_f64 != std::floor(c->goal_reward_f64) || std::fabs(c->goal_reward_f64) > 2147483647.0))
        return fail(RCW_ERR_INVALID_ARGUMENT, "goal_reward_f64 = %g is not an integer the reward type holds", c->goal_reward_f64);
    if (c->world_unit_bits != 32 && c->world_unit_bits != 64)
        return fail(RCW_ERR_INVALID_ARGUMENT, "world_unit_bits must be 32 or 64 (got %d)", c->world_unit_bits);
    if (c->world_unit_bits == 64) {
        if (!(c->player_radius_wu_f64 > 0.0 && c->player_radius_wu_f64 < 0.5))
            return fail(RCW_ERR_INVALID_ARGUMENT, "player_radius_wu_f64 must be in (0, 0.5)");
        if (!(c->position_increment_wu_f64 > 0.0) || !std::isfinite(c->position_increment_wu_f64) ||
            !(c->semi_field_of_view_wu_f64 > 0.0) || !std::isfinite(c->semi_field_of_view_wu_f64) ||
            !(c->camera_height_tile_wu_f64 > 0.0) || !std::isfinite(c->camera_height_tile_wu_f64))
            return fail(RCW_ERR_INVALID_ARGUMENT, "the *_f64 world-unit parameters must be positive and finite");
    }
    if (!(c->player_radius_wu > 0.0f && c->player_radius_wu < 0.5f))   // "should be less than 0.5" SR:47
        return fail(RCW_ERR_INVALID_ARGUMENT, "player_radius_wu must be in (0, 0.5)");
    if (!(c->position_increment_wu > 0.0f) || !std::isfinite(c->position_increment_wu))
        return fail(RCW_ERR_INVALID_ARGUMENT, "position_increment_wu must be positive and finite");
    if (!(c->semi_field_of_view_wu > 0.0f) || !std::isfinite(c->semi_field_of_view_wu))
        return fail(RCW_ERR_INVALID_ARGUMENT, "semi_field_of_view_wu must be positive and finite");
    if (c->render_top_view && (c->pu_per_tu < 1 || c->pu_per_tu > 4096))
        return fail(RCW_ERR_INVALID_ARGUMENT, "pu_per_tu must be in 1..4096 for the top view");
    if (!(c->camera_height_tile_wu > 0.0f) || !std::isfinite(c->camera_height_tile_wu))
        return fail(RCW_ERR_INVALID_ARGUMENT, "camera_height_tile_wu must be positive and finite");
    if (c->dda_tie_break < 0 || c->dda_tie_break > 1 || c->dda_distance < 0 || c->dda_distance > 1 ||
        c->normalize_mode < 0 || c->normalize_mode > 1 || c->out_of_bounds < 0 || c->out_of_bounds > 1)
        return fail(RCW_ERR_INVALID_ARGUMENT, "dda_tie_break / dda_distance / normalize_mode / out_of_bounds out of range");
    return RCW_OK;
}

// The descriptors of the current frames, where the one-launch step left them stale (rcw_handle::cols_live): cast_rays! SR:195-231 on the
// current state, no action — the cast kernel, stream-ordered in front of the reader.
int ensure_columns(rcw_handle* h)
{
    if (!h->cols_stale) return RCW_OK;
    RCW_HIP(rcw_launch_cast(h->dev, nullptr, nullptr, h->stream));
    h->cols_stale = false;
    return RCW_OK;
}

// Wait for the stream, then surface the sticky device error word.
int sync_and_check(rcw_handle* h)
{
    RCW_HIP(hipMemcpyAsync(h->h_err, h->d_err, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    RCW_HIP(hipStreamSynchronize(h->stream));
    const int32_t e = h->h_err[0];
    if (e == RCW_ERR_INVALID_ACTION) return fail(e, "invalid action (must be in 1..%d); the agents it was given to were not stepped (rcw_status)", RCW_NUM_ACTIONS);
    if (e == RCW_ERR_OUT_OF_BOUNDS) return fail(e, "a tile index left the tile map (BoundsError in the reference)");
    if (e == RCW_ERR_HIP) return fail(e, "a kernel gave up waiting for another one after about a second (the top view's store kernel for its draw kernel): the images of that call are not valid");
    if (e != 0) return fail(e, "device error %d", e);
    return RCW_OK;
}

// The Float32 entry points serve Float32 worlds, the *64 ones Float64 worlds.
int check_real(rcw_handle* h, bool want64, const char* fn)
{
    if (h->real64 == want64) return RCW_OK;
    return fail(RCW_ERR_UNSUPPORTED, "%s: the handle's world-unit type is %s; use the %s entry point", fn,
                h->real64 ? "Float64" : "Float32", h->real64 ? "*64" : "Float32");
}

int check_handle(rcw_handle* h)
{
    if (!h) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL handle");
    RCW_HIP(hipSetDevice(h->device));
    return RCW_OK;
}

int upload_mask(rcw_handle* h, const uint8_t* mask_host, const uint8_t** mask_dev)
{
    *mask_dev = nullptr;
    if (!mask_host) return RCW_OK;
    RCW_HIP(hipMemcpyAsync(h->d_mask, mask_host, (size_t)h->B, hipMemcpyHostToDevice, h->stream));
    // the host buffer may be pageable and reused by the caller right away
    RCW_HIP(hipStreamSynchronize(h->stream));
    *mask_dev = (const uint8_t*)h->d_mask;
    return RCW_OK;
}

template <typename T>
int copy_out(rcw_handle* h, T* out_host, const void* dev, size_t count)
{
    if (!out_host) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL output pointer");
    int rc = sync_and_check(h);
    RCW_HIP(hipMemcpy(out_host, dev, count * sizeof(T), hipMemcpyDeviceToHost));
    return rc;
}


// ---- RCCL, loaded on demand ------------------------------------------------------------------------------
// librcw_hip does not link librccl: a single-GPU user never needs it, and inside a process that already
// carries one (PyTorch bundles its own copy under the same soname) a second instance must not be loaded.
// dlopen("librccl.so.1") returns the resident copy when there is one and the system one otherwise.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
RcclApi g_rccl;

int load_rccl()
{
    // several handles of one process (one rank each, a thread each) may get here together: one loads, the others wait
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.lib) return RCW_OK;
    // RCW_RCCL_LIBRARY, where set, is THE library: one that cannot be loaded is an error, not a reason to fall back to another copy
    const char* const chosen = std::getenv("RCW_RCCL_LIBRARY");
    void* lib = nullptr;
    if (chosen && *chosen) {
        lib = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
        if (!lib) return fail(RCW_ERR_UNSUPPORTED, "RCW_RCCL_LIBRARY=%s could not be loaded (%s)", chosen, dlerror());
    } else {
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
    }
    if (!lib) return fail(RCW_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded (%s); set RCW_RCCL_LIBRARY", dlerror());
    RcclApi api;
    api.lib = lib;
#define RCW_SYM(field, name)                                                                       \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(lib, name));                          \
    if (!api.field) { dlclose(lib); return fail(RCW_ERR_UNSUPPORTED, "librccl lacks %s", name); }
    RCW_SYM(GetUniqueId, "ncclGetUniqueId")
    RCW_SYM(CommInitRank, "ncclCommInitRank")
    RCW_SYM(CommDestroy, "ncclCommDestroy")
    RCW_SYM(AllGather, "ncclAllGather")
    RCW_SYM(GroupStart, "ncclGroupStart")
    RCW_SYM(GroupEnd, "ncclGroupEnd")
    RCW_SYM(GetErrorString, "ncclGetErrorString")
    RCW_SYM(GetVersion, "ncclGetVersion")
#undef RCW_SYM
    g_rccl = api;
    return RCW_OK;
}

#define RCW_NCCL(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(RCW_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

int need_comm(rcw_handle* h, const char* fn)
{
    if (!h->comm) return fail(RCW_ERR_INVALID_ARGUMENT, "%s: call rcw_comm_init first", fn);
    return RCW_OK;
}

}  // namespace

extern "C" {

int rcw_abi_version(void) { return RCW_ABI_VERSION; }
const char* rcw_last_error(void) { return g_err; }

int rcw_config_default(rcw_config* c)
{
    if (!c) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL config");
    std::memset(c, 0, sizeof *c);
    c->abi_version = RCW_ABI_VERSION;
    c->height_tile_map_tu = 8;          // SR:260
    c->width_tile_map_tu = 16;          // SR:261
    c->num_directions = 128;            // SR:262
    c->num_rays = 512;                  // SR:268
    c->height_camera_view_pu = 256;     // SR:271
    c->pu_per_tu = 32;                  // SR:269
    c->player_radius_wu = (float)(1.0 / 8.0);        // convert(T, 1/8) SR:263
    c->position_increment_wu = (float)(1.0 / 8.0);   // SR:264
    c->semi_field_of_view_wu = (float)(2.0 / 3.0);   // convert(T, 2/3) SR:267
    c->camera_height_tile_wu = 1.0f;    // SR:270
    c->goal_reward = 1.0f;              // one(R) SR:82
    c->floor_color = 0x00404040u;       // SR:291
    c->ceiling_color = 0x00FFFFFFu;     // SR:292
    c->wall_dim_1_color = 0x00808080u;  // SR:293
    c->wall_dim_2_color = 0x00c0c0c0u;  // SR:294
    c->goal_dim_1_color = 0x00800000u;  // SR:295
    c->goal_dim_2_color = 0x00c00000u;  // SR:296
    c->dda_tie_break = RCW_DDA_TIE_X_FIRST_ON_LT;
    c->dda_distance = RCW_DDA_DIST_SIDE_MINUS_DELTA;
    c->normalize_mode = RCW_NORMALIZE_INV_NORM_TIMES;
    c->auto_reset = 0;
    c->agent_id_offset = 0;
    c->reward_type = RCW_REWARD_FLOAT32;           // R = Float32 SR:266
    c->goal_reward_f64 = 1.0;                      // one(R) SR:82
    c->out_of_bounds = RCW_OOB_ERROR;
    c->world_unit_bits = 32;                       // T = Float32 SR:259
    c->player_radius_wu_f64 = 1.0 / 8.0;           // convert(Float64, .) of the same literals
    c->position_increment_wu_f64 = 1.0 / 8.0;
    c->semi_field_of_view_wu_f64 = 2.0 / 3.0;
    c->camera_height_tile_wu_f64 = 1.0;
    return RCW_OK;
}

int rcw_create(const rcw_config* cfg, int32_t batch, int32_t device, uint64_t seed, rcw_handle** out)
{
    if (!cfg || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    int rc = validate_config(cfg, batch);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(RCW_ERR_NO_DEVICE, "no HIP device visible: librcw_hip has no CPU fallback");
    if (device < 0 || device >= ndev)
        return fail(RCW_ERR_NO_DEVICE, "device %d not in 0..%d", device, ndev - 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(RCW_ERR_NO_DEVICE, "hipGetDeviceProperties(%d) failed", device);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RCW_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    RCW_HIP(hipSetDevice(device));

    rcw_handle* h = new (std::nothrow) rcw_handle();
    if (!h) return fail(RCW_ERR_OUT_OF_MEMORY, "host allocation failed");
    h->cfg = *cfg;
    h->B = batch;
    h->device = device;
    const int H = cfg->height_tile_map_tu, W = cfg->width_tile_map_tu, N = cfg->num_rays;
    const int nd = cfg->num_directions, Hc = cfg->height_camera_view_pu;
    h->nchunks = (2 * H * W + 63) / 64;   // BitArray chunks: cld(2HW, 64)
    h->real64 = cfg->world_unit_bits == 64;
    h->real_size = h->real64 ? sizeof(double) : sizeof(float);
    const size_t B = (size_t)batch;

#define RCW_TRY(expr)                                       \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            free_all(h);                                    \
            delete h;                                       \
            return fail(e_ == hipErrorOutOfMemory ? RCW_ERR_OUT_OF_MEMORY : RCW_ERR_HIP, \
                        "%s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                   \
    } while (0)

    RCW_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    RCW_TRY(hipEventCreate(&h->ev_start));
    RCW_TRY(hipEventCreate(&h->ev_stop));
    RCW_TRY(hipMalloc(&h->d_pos, B * 2 * h->real_size));
    RCW_TRY(hipMalloc(&h->d_dir, B * sizeof(int32_t)));
    RCW_TRY(hipMalloc(&h->d_goal, B * sizeof(int2)));
    h->reward_size = (cfg->reward_type == RCW_REWARD_FLOAT64 || cfg->reward_type == RCW_REWARD_INT64) ? 8 : 4;
    RCW_TRY(hipMalloc(&h->d_reward, B * h->reward_size));
    RCW_TRY(hipMalloc(&h->d_done, B));
    RCW_TRY(hipMalloc(&h->d_episode, B * sizeof(uint32_t)));
    RCW_TRY(hipMalloc(&h->d_tile_map, B * (size_t)h->nchunks * sizeof(uint64_t) + 16));   // (+ 2 words: the flat top store kernel reads three words from any word of a map)
    RCW_TRY(hipMalloc(&h->d_dir_table, (size_t)nd * 2 * h->real_size));
    RCW_TRY(hipMalloc(&h->d_ray_table, (size_t)nd * RCW_TABLE_ROWS * N * h->real_size));
    RCW_TRY(hipMalloc(&h->d_obs, B * (size_t)N * Hc * sizeof(uint32_t)));
    RCW_TRY(hipMalloc(&h->d_col_h, B * (size_t)N * sizeof(int32_t)));
    RCW_TRY(hipMalloc(&h->d_col_c, B * (size_t)N));
    if (cfg->render_top_view)
        RCW_TRY(hipMalloc(&h->d_top_view, B * (size_t)H * W * cfg->pu_per_tu * cfg->pu_per_tu * sizeof(uint32_t)));
    RCW_TRY(hipMalloc(&h->d_err, sizeof(int32_t)));
    RCW_TRY(hipMalloc(&h->d_status, B * sizeof(int32_t)));
    RCW_TRY(hipMalloc(&h->d_actions, B));
    RCW_TRY(hipMalloc(&h->d_mask, B));
    RCW_TRY(hipMalloc(&h->d_in_goal, B * sizeof(int2)));
    RCW_TRY(hipMalloc(&h->d_in_pos, B * 2 * h->real_size));
    RCW_TRY(hipMalloc(&h->d_in_dir, B * sizeof(int32_t)));
    RCW_TRY(hipHostMalloc((void**)&h->h_err, sizeof(int32_t), hipHostMallocDefault));
    for (int k = 0; k < 2; ++k) {
        RCW_TRY(hipHostMalloc((void**)&h->h_actions[k], B, hipHostMallocDefault));
        RCW_TRY(hipEventCreateWithFlags(&h->ev_actions[k], hipEventDisableTiming));
    }
    RCW_TRY(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
    RCW_TRY(hipMemsetAsync(h->d_status, 0, B * sizeof(int32_t), h->stream));
#undef RCW_TRY

    RcwDev& d = h->dev;
    set_geometry(d, cfg, batch);
    d.nwords = h->nchunks * 2;
    d.radius = cfg->player_radius_wu;
    d.radius_sq = cfg->player_radius_wu * cfg->player_radius_wu;        // radius * radius CD:18
    d.inc = cfg->position_increment_wu;
    d.goal_reward = cfg->goal_reward;
    d.goal_reward64 = cfg->goal_reward_f64;
    d.reward_type = cfg->reward_type;
    d.num = cfg->camera_height_tile_wu * (float)N;                      // SR:406 numerator
    d.two_fov = 2.0f * cfg->semi_field_of_view_wu;                      // 2 * fov
    d.radius64 = cfg->player_radius_wu_f64;
    d.radius_sq64 = cfg->player_radius_wu_f64 * cfg->player_radius_wu_f64;
    d.inc64 = cfg->position_increment_wu_f64;
    d.num64 = cfg->camera_height_tile_wu_f64 * (double)N;
    d.two_fov64 = 2.0 * cfg->semi_field_of_view_wu_f64;
    d.floor_color = cfg->floor_color; d.ceiling_color = cfg->ceiling_color;
    d.colour[RCW_COLOUR_WALL_DIM_1] = cfg->wall_dim_1_color;
    d.colour[RCW_COLOUR_WALL_DIM_2] = cfg->wall_dim_2_color;
    d.colour[RCW_COLOUR_GOAL_DIM_1] = cfg->goal_dim_1_color;
    d.colour[RCW_COLOUR_GOAL_DIM_2] = cfg->goal_dim_2_color;
    d.tie_le = cfg->dda_tie_break == RCW_DDA_TIE_X_FIRST_ON_LE;
    d.dist_pre = cfg->dda_distance == RCW_DDA_DIST_PRE_INCREMENT;
    d.auto_reset = cfg->auto_reset ? 1 : 0;
    d.agent_id_offset = cfg->agent_id_offset;
    d.seed = seed;
    d.pos = (float2*)h->d_pos; d.pos64 = (double2*)h->d_pos; d.dir = (int32_t*)h->d_dir; d.goal = (int2*)h->d_goal;
    d.reward = h->d_reward; d.done = (uint8_t*)h->d_done; d.episode = (uint32_t*)h->d_episode;
    d.tile_map = (uint32_t*)h->d_tile_map;
    d.dir_table = (const float2*)h->d_dir_table; d.ray_table = (const float*)h->d_ray_table;
    d.dir_table64 = (const double2*)h->d_dir_table; d.ray_table64 = (const double*)h->d_ray_table;
    d.obs = (uint32_t*)h->d_obs; d.col_h = (int32_t*)h->d_col_h; d.col_c = (uint8_t*)h->d_col_c;
    d.err = (int32_t*)h->d_err;
    d.top_view = (uint32_t*)h->d_top_view;
    d.status = (int32_t*)h->d_status;
    d.oob_empty = cfg->out_of_bounds == RCW_OOB_TREAT_EMPTY;
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->hw.cus = h->num_cus;
    if (prop.sharedMemPerBlock >= 64 * 1024) h->hw.lds_per_cu = (int)prop.sharedMemPerBlock;           // (gfx950: 160 KiB, the whole CU's)
    if (prop.maxThreadsPerMultiProcessor >= 64) h->hw.waves_per_cu = prop.maxThreadsPerMultiProcessor / 64;
    // fill kernel: one workgroup per CU (256 on an MI355X in SPX mode; a partitioned device reports fewer)
    d.fill_grid = h->num_cus; d.fill_plain = 0; d.fill_flat = 0;
    // lanes per agent in the cast kernel: four rays a lane once the batch fills the chip (measured, µs: 512 columns 45.6 vs 51.4
    // with two a lane, 256 columns 12.3 vs 12.7; 1024 columns take 256 lanes either way), two a lane for small batches, where
    // an agent's own latency is what counts
    { const int lanes = h->B >= 1024 ? (N + 3) / 4 : (N + 1) / 2; d.cast_block = lanes >= 256 ? 256 : ((lanes + 63) / 64) * 64; }
    d.cast_ballot = 0; d.cast_table_lds = 0; d.cast_r3 = 0; d.cast_waves = 0; d.top_debug = 0;
    // development builds (make dev: -DRCW_DEV_SWITCHES -> librcw_hip_dev.so) read tuning knobs and the measured-and-rejected
    // kernel variants from the environment; the shipped library reads nothing but RCW_RCCL_LIBRARY
    if (const char* v = RCW_DEV_ENV("RCW_CAST_BLOCK")) { const int b = std::atoi(v); if (b == 64 || b == 128 || b == 192 || b == 256) d.cast_block = b; }
    if (const char* v = RCW_DEV_ENV("RCW_CAST_KERNEL")) d.cast_r3 = std::strcmp(v, "r3") == 0 ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_CAST_WAVES")) d.cast_waves = std::atoi(v) ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_CAST_MARCH")) d.cast_ballot = std::strcmp(v, "ballot") == 0 ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_CAST_TABLE"))   // only where tile bytes + 5 N table values fit the default 64 KiB
        d.cast_table_lds = std::strcmp(v, "lds") == 0 && rcw_step_lds_bytes(d) + 2 * (size_t)H + (size_t)RCW_TABLE_ROWS * N * h->real_size + 128 <= 64 * 1024 ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_FILL_GRID")) { const int g = std::atoi(v); if (g >= 1 && g <= 65536) d.fill_grid = g; }
    if (const char* v = RCW_DEV_ENV("RCW_FILL_PLAIN")) d.fill_plain = std::atoi(v) ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_FILL_FLAT")) d.fill_flat = std::atoi(v) ? 1 : 0;
    if (const char* v = RCW_DEV_ENV("RCW_TOP_DEBUG")) d.top_debug = std::atoi(v);
    d.fill_pairs = 0;
    if (const char* v = RCW_DEV_ENV("RCW_FILL_FLAT_PAIRS")) d.fill_pairs = std::atoi(v);   // 1: two wavefronts a slot; 2: timing only, a prefetch without loads
    d.top_draw_r4 = 0;
    d.top_draw_banks = 0;
    if (const char* v = RCW_DEV_ENV("RCW_TOP_DRAW")) { d.top_draw_r4 = std::strcmp(v, "r4") == 0 ? 1 : 0; d.top_draw_banks = std::strcmp(v, "banks") == 0 ? 1 : (std::strcmp(v, "halfds") == 0 ? 2 : (std::strcmp(v, "nods") == 0 ? 3 : 0)); }
    d.top_rotate = 33;                                                       // (measured: rcw_kernels.hip, rcw_top_store_flat_kernel)
    if (const char* v = RCW_DEV_ENV("RCW_TOP_ROTATE")) { const int r = std::atoi(v); if (r >= 0 && r < 65536) d.top_rotate = r; }
    d.fill_trips = -1;
    if (const char* v = RCW_DEV_ENV("RCW_FILL_TRIPS")) d.fill_trips = std::atoi(v);
    d.step_fused = 0; d.step_flags = nullptr; d.step_hc = nullptr; d.step_epoch = 0;
    if (const char* v = RCW_DEV_ENV("RCW_STEP_FUSED")) {
        if (std::atoi(v)) {
            hipError_t e = hipMalloc(&h->d_step_flags, B * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(&h->d_step_hc, B * (size_t)N * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemset(h->d_step_flags, 0, B * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemset(h->d_step_hc, 0, B * (size_t)N * sizeof(uint32_t));
            if (e != hipSuccess) { free_all(h); delete h; return fail(RCW_ERR_OUT_OF_MEMORY, "fused step: %s", hipGetErrorString(e)); }
            d.step_flags = (uint32_t*)h->d_step_flags; d.step_hc = (uint32_t*)h->d_step_hc; d.step_fused = std::atoi(v) == 2 ? 2 : 1; d.step_epoch = 0;
        }
    }
    if (const char* v = RCW_DEV_ENV("RCW_STEP_PIECES")) {
        h->step_pieces = std::atoi(v) == 2 ? 2 : 1;
        if (h->step_pieces == 2) {                                                            // the side stream and its two events
            hipError_t e = hipStreamCreateWithFlags(&h->top_stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_top_fork, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_top_join[0], hipEventDisableTiming);
            if (e != hipSuccess) { free_all(h); delete h; return fail(RCW_ERR_HIP, "side stream: %s", hipGetErrorString(e)); }
        }
    }
    {
        int want_form = 0, want_runs = 0;
        if (const char* v = RCW_DEV_ENV("RCW_TOP_SPLIT")) { const int f = std::atoi(v); if (!f) want_form = RCW_TOP_VIEW_ONE_KERNEL; else if (f == 2) want_form = RCW_TOP_VIEW_TWO_KERNELS; }
        if (const char* v = RCW_DEV_ENV("RCW_TOP_INPLACE")) { if (std::atoi(v)) want_form = RCW_TOP_VIEW_IN_PLACE; }
        if (const char* v = RCW_DEV_ENV("RCW_TOP_RUNS")) { const int r = std::atoi(v); if (r >= 1 && r <= 8 && r <= batch) want_runs = r; }
        rc = plan_top_view(h, want_form, want_runs, /*lenient=*/true);
        if (rc != RCW_OK) { free_all(h); delete h; return rc; }
    }
    if (rcw_step_lds_bytes(d) > 64 * 1024) {
        free_all(h); delete h;
        return fail(RCW_ERR_UNSUPPORTED, "tile map + column buffer need %zu B of LDS (> 64 KiB)", rcw_step_lds_bytes(d));
    }
#ifdef RCW_DEV_SWITCHES
    d.spec_debug = 0;
    if (const char* v = RCW_DEV_ENV("RCW_SPEC_DEBUG")) d.spec_debug = std::atoi(v);
#endif
    {
        int want = 0;
        if (const char* v = RCW_DEV_ENV("RCW_STEP_FORM")) { const int f = std::atoi(v); if (f == RCW_STEP_TWO_LAUNCHES) want = f; }
        if (d.step_fused || h->step_pieces == 2) want = RCW_STEP_TWO_LAUNCHES;          // (development experiments on the two-launch step)
        rc = plan_step_form(h, want);
        if (rc != RCW_OK) { free_all(h); delete h; return rc; }
    }

    try {
        if (h->real64) build_direction_table<double>(nd, h->dir_table64); else build_direction_table<float>(nd, h->dir_table);
        rebuild_ray_table(h);
    } catch (const std::bad_alloc&) {
        free_all(h); delete h;
        return fail(RCW_ERR_OUT_OF_MEMORY, "host allocation of the (direction, ray) table failed");
    }
    rc = upload_tables(h);
    if (rc == RCW_OK) {
        hipError_t e = rcw_launch_init_tile_map(d, h->stream);
        if (e != hipSuccess) rc = fail(RCW_ERR_HIP, "init_tile_map launch: %s", hipGetErrorString(e));
    }
    if (rc == RCW_OK) rc = rcw_reset(h, nullptr, seed);
    if (rc == RCW_OK) rc = sync_and_check(h);
    if (rc != RCW_OK) { free_all(h); delete h; return rc; }
    *out = h;
    return RCW_OK;
}

int rcw_destroy(rcw_handle* h)
{
    if (!h) return RCW_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
    free_all(h);
    delete h;
    return RCW_OK;
}

extern "C++" {
template <typename T>
int set_direction_table_impl(rcw_handle* h, const T* directions_wu, std::vector<T>& table)
{
    if (!directions_wu) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL direction table");
    RCW_HIP(hipStreamSynchronize(h->stream));
    try {
        table.assign(directions_wu, directions_wu + (size_t)2 * h->cfg.num_directions);
        rebuild_ray_table(h);
    } catch (const std::bad_alloc&) {
        return fail(RCW_ERR_OUT_OF_MEMORY, "host allocation of the (direction, ray) table failed");
    }
    int rc = upload_tables(h); if (rc) return rc;
    RCW_HIP(launch_step(h, nullptr, nullptr));   // re-render
    return RCW_OK;
}
}  // extern "C++"

int rcw_set_direction_table(rcw_handle* h, const float* directions_wu)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, false, "rcw_set_direction_table"); if (rc) return rc;
    return set_direction_table_impl<float>(h, directions_wu, h->dir_table);
}
int rcw_set_direction_table64(rcw_handle* h, const double* directions_wu)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, true, "rcw_set_direction_table64"); if (rc) return rc;
    return set_direction_table_impl<double>(h, directions_wu, h->dir_table64);
}

int rcw_set_stream(rcw_handle* h, void* hip_stream)
{
    int rc = check_handle(h); if (rc) return rc;
    RCW_HIP(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return RCW_OK;
}

int rcw_get_stream(rcw_handle* h, void** hip_stream)
{
    if (!h || !hip_stream) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *hip_stream = (void*)h->stream;
    return RCW_OK;
}

int rcw_bind_obs(rcw_handle* h, void* device_ptr)
{
    int rc = check_handle(h); if (rc) return rc;
    if (device_ptr && ((uintptr_t)device_ptr & 15u))
        return fail(RCW_ERR_INVALID_ARGUMENT, "observation buffer must be 16-byte aligned");
    // No synchronisation: the pointer travels in the kernel arguments of the launches that follow,
    // work already enqueued keeps the buffer it was launched with (double-buffered observations).
    h->dev.obs = device_ptr ? (uint32_t*)device_ptr : (uint32_t*)h->d_obs;
    return RCW_OK;
}

int rcw_reset(rcw_handle* h, const uint8_t* mask_host, uint64_t seed)
{
    int rc = check_handle(h); if (rc) return rc;
    const uint8_t* mask_dev = nullptr;
    rc = upload_mask(h, mask_host, &mask_dev); if (rc) return rc;
    // (the seed is the HANDLE's: an agent that is done under auto_reset and NOT in the mask is re-sampled by its next action with the new
    // seed — but the one-launch step has already cast that agent's successors from a preview drawn with the old one: every agent's slots are
    // cast again by the next step, as a launch of its own)
    if (seed != h->dev.seed && h->dev.auto_reset && mask_dev) h->spec_primed = false;
    h->dev.seed = seed;
    RCW_HIP(rcw_launch_reset(h->dev, mask_dev, h->stream));            // SR:110-132
    RCW_HIP(launch_step(h, nullptr, mask_dev));    // SR:134, SR:329
    return RCW_OK;
}

extern "C++" {
template <typename T>
int set_state_impl(rcw_handle* h, const int32_t* goal_ij, const T* position_wu, const int32_t* direction_au,
                   const uint8_t* mask_host)
{
    if (!goal_ij || !position_wu || !direction_au) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL state array");
    const int H = h->cfg.height_tile_map_tu, W = h->cfg.width_tile_map_tu;
    for (int32_t a = 0; a < h->B; ++a) {
        if (mask_host && !mask_host[a]) continue;
        const int gi = goal_ij[2 * a], gj = goal_ij[2 * a + 1];
        if (gi < 2 || gi > H - 1 || gj < 2 || gj > W - 1)   // rand(2:H-1), rand(2:W-1) SR:120
            return fail(RCW_ERR_INVALID_ARGUMENT, "agent %d: goal (%d,%d) not an interior tile", a, gi, gj);
        if (direction_au[a] < 0 || direction_au[a] >= h->cfg.num_directions)
            return fail(RCW_ERR_INVALID_ARGUMENT, "agent %d: direction %d not in 0..%d", a, direction_au[a], h->cfg.num_directions - 1);
        const T x = position_wu[2 * a], y = position_wu[2 * a + 1];
        if (!(std::isfinite(x) && std::isfinite(y) && x >= (T)1 && x < (T)(H - 1) && y >= (T)1 && y < (T)(W - 1)))
            return fail(RCW_ERR_INVALID_ARGUMENT, "agent %d: position (%g,%g) not inside the room", a, (double)x, (double)y);
    }
    const uint8_t* mask_dev = nullptr;
    int rc = upload_mask(h, mask_host, &mask_dev); if (rc) return rc;
    const size_t B = (size_t)h->B;
    RCW_HIP(hipMemcpyAsync(h->d_in_goal, goal_ij, B * sizeof(int2), hipMemcpyHostToDevice, h->stream));
    RCW_HIP(hipMemcpyAsync(h->d_in_pos, position_wu, B * 2 * sizeof(T), hipMemcpyHostToDevice, h->stream));
    RCW_HIP(hipMemcpyAsync(h->d_in_dir, direction_au, B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    RCW_HIP(hipStreamSynchronize(h->stream));
    RCW_HIP(rcw_launch_set_state(h->dev, (const int2*)h->d_in_goal, h->d_in_pos, (const int32_t*)h->d_in_dir, mask_dev,
                                 h->stream));
    RCW_HIP(launch_step(h, nullptr, mask_dev));
    return RCW_OK;
}
}  // extern "C++"

int rcw_set_state(rcw_handle* h, const int32_t* goal_ij, const float* position_wu,
                  const int32_t* direction_au, const uint8_t* mask_host)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, false, "rcw_set_state"); if (rc) return rc;
    return set_state_impl<float>(h, goal_ij, position_wu, direction_au, mask_host);
}
int rcw_set_state64(rcw_handle* h, const int32_t* goal_ij, const double* position_wu,
                    const int32_t* direction_au, const uint8_t* mask_host)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, true, "rcw_set_state64"); if (rc) return rc;
    return set_state_impl<double>(h, goal_ij, position_wu, direction_au, mask_host);
}

int rcw_step(rcw_handle* h, const uint8_t* actions_host)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!actions_host) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL actions");
    for (int32_t a = 0; a < h->B; ++a)   // @assert action in Base.OneTo(NUM_ACTIONS) SR:140
        if (actions_host[a] < 1 || actions_host[a] > RCW_NUM_ACTIONS)
            return fail(RCW_ERR_INVALID_ACTION, "Invalid action: %d (agent %d)", (int)actions_host[a], a);
    // Stage through a pinned ring so the caller may reuse its buffer at once and the host
    // can run one step ahead of the GPU; the copy is ordered on the stream behind the
    // previous step, which is still reading d_actions.
    const int slot = h->action_slot;
    h->action_slot ^= 1;
    RCW_HIP(hipEventSynchronize(h->ev_actions[slot]));
    std::memcpy(h->h_actions[slot], actions_host, (size_t)h->B);
    RCW_HIP(hipMemcpyAsync(h->d_actions, h->h_actions[slot], (size_t)h->B, hipMemcpyHostToDevice, h->stream));
    RCW_HIP(hipEventRecord(h->ev_actions[slot], h->stream));
    RCW_HIP(launch_step(h, (const uint8_t*)h->d_actions, nullptr));
    return RCW_OK;
}

int rcw_step_device(rcw_handle* h, const uint8_t* actions_device)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!actions_device) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL actions");
    RCW_HIP(launch_step(h, actions_device, nullptr));
    return RCW_OK;
}

int rcw_cast_rays(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    RCW_HIP(rcw_launch_cast(h->dev, nullptr, nullptr, h->stream));   // no action: rays + descriptors only
    h->cols_stale = false;
    return RCW_OK;
}

int rcw_update_camera_view(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = ensure_columns(h); if (rc) return rc;
    RCW_HIP(rcw_launch_fill(h->dev, h->dev.col_h, h->dev.col_c, h->dev.obs, (long long)h->dev.B * h->dev.N, nullptr, h->stream));
    return RCW_OK;
}

int rcw_update_top_view(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!h->d_top_view) return fail(RCW_ERR_UNSUPPORTED, "handle was created with render_top_view = 0");
    RCW_HIP(launch_top_view(h, nullptr, false, [](hipStream_t) { return hipSuccess; }));
    return RCW_OK;
}

int rcw_sync(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    return sync_and_check(h);
}

int rcw_clear_error(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    RCW_HIP(hipMemsetAsync(h->d_err, 0, sizeof(int32_t), h->stream));
    RCW_HIP(hipMemsetAsync(h->d_status, 0, (size_t)h->B * sizeof(int32_t), h->stream));
    RCW_HIP(hipStreamSynchronize(h->stream));
    return RCW_OK;
}

int rcw_obs_device_ptr(rcw_handle* h, void** device_ptr)
{
    if (!h || !device_ptr) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *device_ptr = h->dev.obs;
    return RCW_OK;
}

int rcw_obs_copy(rcw_handle* h, uint32_t* out_host, int32_t first, int32_t count)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!out_host || first < 0 || count < 0 || first + (int64_t)count > h->B)
        return fail(RCW_ERR_INVALID_ARGUMENT, "bad agent range [%d, %d)", first, first + count);
    rc = sync_and_check(h);
    const size_t frame = (size_t)h->cfg.num_rays * h->cfg.height_camera_view_pu;
    RCW_HIP(hipMemcpy(out_host, h->dev.obs + (size_t)first * frame, (size_t)count * frame * sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    return rc;
}

int rcw_top_view_device_ptr(rcw_handle* h, void** device_ptr)
{
    if (!h || !device_ptr) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!h->d_top_view) return fail(RCW_ERR_UNSUPPORTED, "handle was created with render_top_view = 0");
    *device_ptr = h->d_top_view;
    return RCW_OK;
}

int rcw_top_view_copy(rcw_handle* h, uint32_t* out_host, int32_t first, int32_t count)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!h->d_top_view) return fail(RCW_ERR_UNSUPPORTED, "handle was created with render_top_view = 0");
    if (!out_host || first < 0 || count < 0 || first + (int64_t)count > h->B)
        return fail(RCW_ERR_INVALID_ARGUMENT, "bad agent range [%d, %d)", first, first + count);
    rc = sync_and_check(h);
    const size_t frame = (size_t)h->cfg.height_tile_map_tu * h->cfg.width_tile_map_tu * h->cfg.pu_per_tu * h->cfg.pu_per_tu;
    RCW_HIP(hipMemcpy(out_host, (uint32_t*)h->d_top_view + (size_t)first * frame, (size_t)count * frame * sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    return rc;
}

int rcw_reward(rcw_handle* h, float* out)
{
    int rc = check_handle(h); if (rc) return rc;
    if (h->cfg.reward_type != RCW_REWARD_FLOAT32)
        return fail(RCW_ERR_UNSUPPORTED, "rcw_reward: the handle's reward type is not Float32; use rcw_reward_typed");
    return copy_out(h, out, h->d_reward, (size_t)h->B);
}
int rcw_reward_typed(rcw_handle* h, void* out)
{
    int rc = check_handle(h); if (rc) return rc;
    return copy_out(h, static_cast<uint8_t*>(out), h->d_reward, (size_t)h->B * h->reward_size);
}
int rcw_done(rcw_handle* h, uint8_t* out) { int rc = check_handle(h); if (rc) return rc; return copy_out(h, out, h->d_done, (size_t)h->B); }
int rcw_position(rcw_handle* h, float* out)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, false, "rcw_position"); if (rc) return rc;
    return copy_out(h, out, h->d_pos, (size_t)2 * h->B);
}
int rcw_position64(rcw_handle* h, double* out)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, true, "rcw_position64"); if (rc) return rc;
    return copy_out(h, out, h->d_pos, (size_t)2 * h->B);
}
int rcw_direction(rcw_handle* h, int32_t* out) { int rc = check_handle(h); if (rc) return rc; return copy_out(h, out, h->d_dir, (size_t)h->B); }
int rcw_goal(rcw_handle* h, int32_t* out) { int rc = check_handle(h); if (rc) return rc; return copy_out(h, out, h->d_goal, (size_t)2 * h->B); }
int rcw_episode(rcw_handle* h, uint32_t* out) { int rc = check_handle(h); if (rc) return rc; return copy_out(h, out, h->d_episode, (size_t)h->B); }

int rcw_status(rcw_handle* h, int32_t* out)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL output pointer");
    RCW_HIP(hipStreamSynchronize(h->stream));
    RCW_HIP(hipMemcpy(out, h->d_status, (size_t)h->B * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RCW_OK;
}

int rcw_reward_device_ptr(rcw_handle* h, void** p)
{
    if (!h || !p) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *p = h->d_reward; return RCW_OK;
}
int rcw_done_device_ptr(rcw_handle* h, void** p)
{
    if (!h || !p) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *p = h->d_done; return RCW_OK;
}

int rcw_tile_map_num_chunks(rcw_handle* h, int32_t* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = h->nchunks; return RCW_OK;
}
int rcw_tile_map_chunks(rcw_handle* h, uint64_t* out)
{
    int rc = check_handle(h); if (rc) return rc;
    return copy_out(h, out, h->d_tile_map, (size_t)h->nchunks * h->B);
}

extern "C++" {
template <typename T>
int rays_impl(rcw_handle* h, int32_t first, int32_t count, int64_t* stop_ij, int64_t* hit_dimension,
              T* distance_wu, T* directions_wu)
{
    if (first < 0 || count < 1 || first + (int64_t)count > h->B)
        return fail(RCW_ERR_INVALID_ARGUMENT, "bad agent range [%d, %d)", first, first + count);
    const size_t n = (size_t)count * h->cfg.num_rays;
    RcwRayOut out{};
    // device scratch lives in the handle and only ever grows: no hipMalloc/hipFree per call
    const size_t want[4] = {stop_ij ? 2 * n * sizeof(int64_t) : 0, hit_dimension ? n * sizeof(int64_t) : 0,
                            distance_wu ? n * sizeof(T) : 0, directions_wu ? 2 * n * sizeof(T) : 0};
    for (int k = 0; k < 4; ++k) {
        if (want[k] <= h->rays_cap[k]) continue;
        if (h->d_rays[k]) { RCW_HIP(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_rays[k]); h->d_rays[k] = nullptr; h->rays_cap[k] = 0; }
        RCW_HIP(hipMalloc(&h->d_rays[k], want[k]));
        h->rays_cap[k] = want[k];
    }
    if (stop_ij) out.stop_ij = (int64_t*)h->d_rays[0];
    if (hit_dimension) out.hit_dim = (int64_t*)h->d_rays[1];
    if (distance_wu) out.dist = h->d_rays[2];
    if (directions_wu) out.dirs = h->d_rays[3];
    RCW_HIP(rcw_launch_rays(h->dev, first, count, out, h->stream));
    RCW_HIP(hipStreamSynchronize(h->stream));
    if (stop_ij) RCW_HIP(hipMemcpy(stop_ij, h->d_rays[0], want[0], hipMemcpyDeviceToHost));
    if (hit_dimension) RCW_HIP(hipMemcpy(hit_dimension, h->d_rays[1], want[1], hipMemcpyDeviceToHost));
    if (distance_wu) RCW_HIP(hipMemcpy(distance_wu, h->d_rays[2], want[2], hipMemcpyDeviceToHost));
    if (directions_wu) RCW_HIP(hipMemcpy(directions_wu, h->d_rays[3], want[3], hipMemcpyDeviceToHost));
    return RCW_OK;
}
}  // extern "C++"

int rcw_rays(rcw_handle* h, int32_t first, int32_t count, int64_t* stop_ij, int64_t* hit_dimension,
             float* distance_wu, float* directions_wu)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, false, "rcw_rays"); if (rc) return rc;
    return rays_impl<float>(h, first, count, stop_ij, hit_dimension, distance_wu, directions_wu);
}
int rcw_rays64(rcw_handle* h, int32_t first, int32_t count, int64_t* stop_ij, int64_t* hit_dimension,
               double* distance_wu, double* directions_wu)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = check_real(h, true, "rcw_rays64"); if (rc) return rc;
    return rays_impl<double>(h, first, count, stop_ij, hit_dimension, distance_wu, directions_wu);
}

int rcw_columns(rcw_handle* h, int32_t first, int32_t count, int32_t* height_line_pu, uint8_t* colour_id)
{
    int rc = check_handle(h); if (rc) return rc;
    if (first < 0 || count < 0 || first + (int64_t)count > h->B)
        return fail(RCW_ERR_INVALID_ARGUMENT, "bad agent range [%d, %d)", first, first + count);
    rc = ensure_columns(h); if (rc) return rc;
    rc = sync_and_check(h);
    const size_t N = (size_t)h->cfg.num_rays;
    if (height_line_pu)
        RCW_HIP(hipMemcpy(height_line_pu, (int32_t*)h->d_col_h + (size_t)first * N, (size_t)count * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (colour_id)
        RCW_HIP(hipMemcpy(colour_id, (uint8_t*)h->d_col_c + (size_t)first * N, (size_t)count * N, hipMemcpyDeviceToHost));
    return rc;
}

int rcw_columns_device_ptr(rcw_handle* h, void** height_line_pu, void** colour_id)
{
    if (!h) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL handle");
    {   // from now on every step refreshes the descriptors (the caller reads them through the pointers, behind the library's back)
        int rc = check_handle(h); if (rc) return rc;
        h->cols_live = true;
        rc = ensure_columns(h); if (rc) return rc;
    }
    if (height_line_pu) *height_line_pu = h->d_col_h;
    if (colour_id) *colour_id = h->d_col_c;
    return RCW_OK;
}

int rcw_expand_columns(rcw_handle* h, const int32_t* height_line_pu_device, const uint8_t* colour_id_device,
                       int32_t count, void* frames_device)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!height_line_pu_device || !colour_id_device || !frames_device || count < 1)
        return fail(RCW_ERR_INVALID_ARGUMENT, "bad argument");
    if ((uintptr_t)frames_device & 15u) return fail(RCW_ERR_INVALID_ARGUMENT, "frames must be 16-byte aligned");
    RCW_HIP(rcw_launch_expand(h->dev, height_line_pu_device, colour_id_device, count, (uint32_t*)frames_device, h->stream));
    return RCW_OK;
}

// ---- the observation gather (RCCL over xGMI) --------------------------------------------------------------
int rcw_comm_unique_id(void* out_id)
{
    if (!out_id) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = load_rccl(); if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == RCW_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    RCW_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof id);
    return RCW_OK;
}

int rcw_comm_init(rcw_handle* h, const void* unique_id, int32_t rank, int32_t world)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!unique_id || world < 1 || rank < 0 || rank >= world)
        return fail(RCW_ERR_INVALID_ARGUMENT, "bad rank %d / world %d", rank, world);
    if (h->comm) return fail(RCW_ERR_INVALID_ARGUMENT, "the handle already has a communicator (rcw_comm_destroy first)");
    rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    RCW_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
    h->comm = comm; h->comm_rank = rank; h->comm_world = world;
    return RCW_OK;
}

int rcw_comm_destroy(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!h->comm) return RCW_OK;
    RCW_HIP(hipStreamSynchronize(h->stream));
    RCW_NCCL(g_rccl.CommDestroy((ncclComm_t)h->comm));
    h->comm = nullptr; h->comm_rank = 0; h->comm_world = 0;
    // the gathered-descriptor scratch is sized by the world: a later rcw_comm_init may have another
    if (h->d_gather_h) (void)hipFree(h->d_gather_h);
    if (h->d_gather_c) (void)hipFree(h->d_gather_c);
    h->d_gather_h = h->d_gather_c = nullptr;
    return RCW_OK;
}

int rcw_comm_info(rcw_handle* h, int32_t* rank, int32_t* world)
{
    if (!h || !rank || !world) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *rank = h->comm_rank; *world = h->comm_world;
    return RCW_OK;
}

int rcw_gather_columns(rcw_handle* h, int32_t* height_all, uint8_t* colour_all)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = need_comm(h, "rcw_gather_columns"); if (rc) return rc;
    if (!height_all || !colour_all) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    rc = ensure_columns(h); if (rc) return rc;
    const size_t n = (size_t)h->B * h->cfg.num_rays;
    // one fused group: the two all-gathers progress together on the handle's stream, behind the step
    RCW_NCCL(g_rccl.GroupStart());
    ncclResult_t r1 = g_rccl.AllGather(h->d_col_h, height_all, n, ncclInt32, (ncclComm_t)h->comm, h->stream);
    ncclResult_t r2 = g_rccl.AllGather(h->d_col_c, colour_all, n, ncclUint8, (ncclComm_t)h->comm, h->stream);
    RCW_NCCL(g_rccl.GroupEnd());
    RCW_NCCL(r1); RCW_NCCL(r2);
    return RCW_OK;
}

int rcw_gather_observations(rcw_handle* h, int32_t mode, void* frames_all)
{
    int rc = check_handle(h); if (rc) return rc;
    rc = need_comm(h, "rcw_gather_observations"); if (rc) return rc;
    if (!frames_all) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    if ((uintptr_t)frames_all & 15u) return fail(RCW_ERR_INVALID_ARGUMENT, "frames must be 16-byte aligned");
    const size_t N = (size_t)h->cfg.num_rays, Hc = (size_t)h->cfg.height_camera_view_pu;
    if (mode == RCW_GATHER_FRAMES) {
        RCW_NCCL(g_rccl.AllGather(h->dev.obs, frames_all, (size_t)h->B * N * Hc, ncclUint32, (ncclComm_t)h->comm, h->stream));
        return RCW_OK;
    }
    if (mode != RCW_GATHER_COLUMNS) return fail(RCW_ERR_INVALID_ARGUMENT, "unknown gather mode %d", mode);
    const size_t all = (size_t)h->B * h->comm_world;
    if ((long long)all > 0x7fffffffll) return fail(RCW_ERR_UNSUPPORTED, "global batch too large");
    if (!h->d_gather_h) RCW_HIP(hipMalloc(&h->d_gather_h, all * N * sizeof(int32_t)));
    if (!h->d_gather_c) RCW_HIP(hipMalloc(&h->d_gather_c, all * N));
    rc = rcw_gather_columns(h, (int32_t*)h->d_gather_h, (uint8_t*)h->d_gather_c); if (rc) return rc;
    RCW_HIP(rcw_launch_expand(h->dev, (const int32_t*)h->d_gather_h, (const uint8_t*)h->d_gather_c, (int32_t)all,
                              (uint32_t*)frames_all, h->stream));
    return RCW_OK;
}

int rcw_device_malloc(rcw_handle* h, uint64_t bytes, void** device_ptr)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!device_ptr || bytes == 0) return fail(RCW_ERR_INVALID_ARGUMENT, "bad argument");
    *device_ptr = nullptr;
    RCW_HIP(hipMalloc(device_ptr, (size_t)bytes));
    return RCW_OK;
}
int rcw_device_free(rcw_handle* h, void* device_ptr)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!device_ptr) return RCW_OK;
    RCW_HIP(hipStreamSynchronize(h->stream));   // work enqueued on the handle may still use it
    RCW_HIP(hipFree(device_ptr));
    return RCW_OK;
}
int rcw_memcpy_to_host(rcw_handle* h, void* dst_host, const void* src_device, uint64_t bytes)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!dst_host || !src_device) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    rc = sync_and_check(h);
    RCW_HIP(hipMemcpy(dst_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost));
    return rc;
}

int rcw_ray_table(rcw_handle* h, float* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_real(h, false, "rcw_ray_table"); if (rc) return rc;
    std::memcpy(out, h->ray_table.data(), h->ray_table.size() * sizeof(float));
    return RCW_OK;
}
int rcw_direction_table(rcw_handle* h, float* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_real(h, false, "rcw_direction_table"); if (rc) return rc;
    std::memcpy(out, h->dir_table.data(), h->dir_table.size() * sizeof(float));
    return RCW_OK;
}
int rcw_ray_table64(rcw_handle* h, double* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_real(h, true, "rcw_ray_table64"); if (rc) return rc;
    std::memcpy(out, h->ray_table64.data(), h->ray_table64.size() * sizeof(double));
    return RCW_OK;
}
int rcw_direction_table64(rcw_handle* h, double* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    int rc = check_real(h, true, "rcw_direction_table64"); if (rc) return rc;
    std::memcpy(out, h->dir_table64.data(), h->dir_table64.size() * sizeof(double));
    return RCW_OK;
}

int rcw_timer_start(rcw_handle* h)
{
    int rc = check_handle(h); if (rc) return rc;
    RCW_HIP(hipEventRecord(h->ev_start, h->stream));
    return RCW_OK;
}
int rcw_timer_stop(rcw_handle* h, float* elapsed_ms)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!elapsed_ms) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    RCW_HIP(hipEventRecord(h->ev_stop, h->stream));
    RCW_HIP(hipEventSynchronize(h->ev_stop));
    RCW_HIP(hipEventElapsedTime(elapsed_ms, h->ev_start, h->ev_stop));
    return RCW_OK;
}

int rcw_profile(rcw_handle* h, int32_t enable)
{
    int rc = check_handle(h); if (rc) return rc;
    RCW_HIP(hipStreamSynchronize(h->stream));
    if (enable && h->prof_ev.empty()) {
        h->prof_ev.resize(4 * kProfileSlots, nullptr);
        for (auto& ev : h->prof_ev) RCW_HIP(hipEventCreate(&ev));
    }
    h->profiling = enable != 0;
    h->prof_count = 0;
    return RCW_OK;
}

int rcw_profile_read(rcw_handle* h, float* cast_ms, float* top_view_ms, float* fill_ms, int32_t* steps)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!cast_ms || !top_view_ms || !fill_ms || !steps) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    RCW_HIP(hipStreamSynchronize(h->stream));
    double c = 0.0, t = 0.0, f = 0.0;
    for (int k = 0; k < h->prof_count; ++k) {
        float a = 0.0f, b = 0.0f, d = 0.0f;
        RCW_HIP(hipEventElapsedTime(&a, h->prof_ev[4 * k], h->prof_ev[4 * k + 1]));
        RCW_HIP(hipEventElapsedTime(&b, h->prof_ev[4 * k + 1], h->prof_ev[4 * k + 2]));
        RCW_HIP(hipEventElapsedTime(&d, h->prof_ev[4 * k + 2], h->prof_ev[4 * k + 3]));
        c += a;
        if (h->dev.top_split) { f += b; t += d; } else { t += b; f += d; }      // two-kernel top view: cast | fill (+ draw beside it) | store
    }
    *steps = h->prof_count;
    *cast_ms = h->prof_count ? (float)(c / h->prof_count) : 0.0f;
    *top_view_ms = h->prof_count && h->dev.top_view ? (float)(t / h->prof_count) : 0.0f;
    *fill_ms = h->prof_count ? (float)(f / h->prof_count) : 0.0f;
    return RCW_OK;
}

int rcw_top_view_form(rcw_handle* h, int32_t* form)
{
    if (!h || !form) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    const RcwDev& d = h->dev;
    *form = !d.top_view ? RCW_TOP_VIEW_NONE : d.top_split ? RCW_TOP_VIEW_TWO_KERNELS : d.top_lds ? RCW_TOP_VIEW_ONE_KERNEL : RCW_TOP_VIEW_IN_PLACE;
    return RCW_OK;
}

int rcw_update_top_view_form(rcw_handle* h, int32_t* form)
{
    if (!h || !form) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    const RcwDev& d = h->dev;
    *form = !d.top_view ? RCW_TOP_VIEW_NONE : (d.top_split && d.top_alone_split) ? RCW_TOP_VIEW_TWO_KERNELS : d.top_lds ? RCW_TOP_VIEW_ONE_KERNEL : RCW_TOP_VIEW_IN_PLACE;
    return RCW_OK;
}

int rcw_set_top_view_form(rcw_handle* h, int32_t form, int32_t runs)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!h->d_top_view) return fail(RCW_ERR_UNSUPPORTED, "handle was created with render_top_view = 0");
    if (form != 0 && form != RCW_TOP_VIEW_IN_PLACE && form != RCW_TOP_VIEW_ONE_KERNEL && form != RCW_TOP_VIEW_TWO_KERNELS)
        return fail(RCW_ERR_INVALID_ARGUMENT, "form must be 0 (automatic) or RCW_TOP_VIEW_IN_PLACE / ONE_KERNEL / TWO_KERNELS (got %d)", form);
    if (runs < 0 || runs > 8) return fail(RCW_ERR_INVALID_ARGUMENT, "runs must be 0 (automatic) or 1..8 (got %d)", runs);
    RCW_HIP(hipStreamSynchronize(h->stream));        // the scratch of the current form may be in use
    rc = plan_top_view(h, form, runs, /*lenient=*/false);
    if (rc != RCW_OK) {                               // leave a usable handle behind: back to the automatic choice
        const int rc2 = plan_top_view(h, 0, 0, true);
        return rc2 != RCW_OK ? rc2 : rc;
    }
    return RCW_OK;
}

int rcw_step_form(rcw_handle* h, int32_t* form)
{
    if (!h || !form) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *form = h->spec_on ? RCW_STEP_ONE_LAUNCH : RCW_STEP_TWO_LAUNCHES;
    return RCW_OK;
}

int rcw_set_step_form(rcw_handle* h, int32_t form)
{
    int rc = check_handle(h); if (rc) return rc;
    if (form != 0 && form != RCW_STEP_TWO_LAUNCHES && form != RCW_STEP_ONE_LAUNCH)
        return fail(RCW_ERR_INVALID_ARGUMENT, "form must be 0 (automatic) or RCW_STEP_TWO_LAUNCHES / RCW_STEP_ONE_LAUNCH (got %d)", form);
    const bool was_on = h->spec_on != 0;
    rc = plan_step_form(h, form); if (rc) return rc;
    if (h->spec_on && !was_on) RCW_HIP(launch_step(h, nullptr, nullptr));   // prime the slots (re-renders the current frames: the same pixels)
    return RCW_OK;
}

int rcw_fill_kernel_name(rcw_handle* h, char* buf, int32_t buflen)
{
    if (!h || !buf || buflen < 1) return fail(RCW_ERR_INVALID_ARGUMENT, "bad argument");
    std::snprintf(buf, (size_t)buflen, "%s", h->spec_on ? (h->dev.Hc == 256 ? "rcw_fill256_cast_kernel" : "rcw_fill_window_cast_kernel") : rcw_fill_kernel_name(h->dev, (long long)h->dev.B * h->dev.N));
    return RCW_OK;
}

int rcw_batch(rcw_handle* h, int32_t* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = h->B; return RCW_OK;
}
int rcw_get_config(rcw_handle* h, rcw_config* out)
{
    if (!h || !out) return fail(RCW_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = h->cfg; return RCW_OK;
}
int rcw_device_name(rcw_handle* h, char* buf, int32_t buflen)
{
    int rc = check_handle(h); if (rc) return rc;
    if (!buf || buflen < 1) return fail(RCW_ERR_INVALID_ARGUMENT, "bad buffer");
    hipDeviceProp_t prop;
    RCW_HIP(hipGetDeviceProperties(&prop, h->device));
    std::snprintf(buf, (size_t)buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return RCW_OK;
}

}  // extern "C"
