/* Plain-C caller of librcw_hip.so, making the calls julia/BatchedSingleRoom.jl makes through
 * ccall (same argument types, host buffers in, host buffers out): test/runtests.jl:15-44 for a
 * batch, then a deterministic rollout whose frame checksum the pytest wrapper compares with the
 * CPU oracle.  No Python, no torch: only include/rcw.h.
 *
 *   gcc -O2 -I include tests/c_abi_harness.c -o harness -L raycastworlds.jl_amd/lib -lrcw_hip
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rcw.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != RCW_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rcw_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static uint64_t lcg(uint64_t* s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return *s >> 33; }

int main(int argc, char** argv)
{
    const int B = 64, STEPS = argc > 1 ? atoi(argv[1]) : 120;
    rcw_config cfg;
    CHECK(rcw_config_default(&cfg));
    cfg.height_tile_map_tu = 8; cfg.width_tile_map_tu = 8; cfg.num_rays = 64;
    cfg.out_of_bounds = RCW_OOB_TREAT_EMPTY;
    rcw_handle* h = NULL;
    CHECK(rcw_create(&cfg, B, 0, 2024, &h));

    float reward[64]; uint8_t done[64], actions[64];
    CHECK(rcw_reward(h, reward)); CHECK(rcw_done(h, done));
    for (int a = 0; a < B; ++a)
        if (reward[a] != 0.0f || done[a]) { fprintf(stderr, "after reset: reward/done not clear\n"); return 1; }

    /* the @assert of single_room.jl:140 */
    memset(actions, 1, sizeof actions); actions[7] = 5;
    if (rcw_step(h, actions) != RCW_ERR_INVALID_ACTION) { fprintf(stderr, "invalid action accepted\n"); return 1; }

    /* a step's two forms (rcw_step_form): two launches by the rule here (64 agents: too small a batch for the one-launch step to pay); the
     * one-launch form on request for a stretch in the middle — the same state and pixels either way (the checksum below is the oracle's) */
    int32_t step_form = -1;
    CHECK(rcw_step_form(h, &step_form));
    if (step_form != RCW_STEP_TWO_LAUNCHES) { fprintf(stderr, "rcw_step_form: %d, expected the two-launch step at this batch\n", (int)step_form); return 1; }
    printf("step_form=%d\n", (int)step_form);

    uint64_t seed = 99;
    float ret[64] = {0};
    int finished = 0;
    for (int s = 0; s < STEPS; ++s) {
        if (s == STEPS / 3) { CHECK(rcw_set_step_form(h, RCW_STEP_ONE_LAUNCH)); CHECK(rcw_step_form(h, &step_form)); if (step_form != RCW_STEP_ONE_LAUNCH) return 1; }
        if (s == 2 * STEPS / 3) CHECK(rcw_set_step_form(h, 0));
        for (int a = 0; a < B; ++a) actions[a] = (uint8_t)(1 + lcg(&seed) % 4);
        CHECK(rcw_step(h, actions));
        CHECK(rcw_reward(h, reward)); CHECK(rcw_done(h, done));
        for (int a = 0; a < B; ++a) {
            if (!done[a] && reward[a] != 0.0f) { fprintf(stderr, "reward on a non-terminal step\n"); return 1; }
            if (done[a] && reward[a] != 1.0f) { fprintf(stderr, "terminal step without goal_reward\n"); return 1; }
            ret[a] += reward[a]; finished += done[a];
        }
    }
    const size_t npix = (size_t)B * 64 * 256;
    uint32_t* frames = (uint32_t*)malloc(npix * sizeof(uint32_t));
    CHECK(rcw_obs_copy(h, frames, 0, B));
    uint64_t sum = 1469598103934665603ULL;                 /* FNV-1a over the words */
    for (size_t k = 0; k < npix; ++k) { sum ^= frames[k]; sum *= 1099511628211ULL; }
    float pos[128]; int32_t dir[64];
    CHECK(rcw_position(h, pos)); CHECK(rcw_direction(h, dir));
    char name[128];
    CHECK(rcw_device_name(h, name, sizeof name));
    printf("device=%s steps=%d terminal_events=%d checksum=%016llx pos0=%.9g,%.9g dir0=%d\n", name, STEPS, finished,
           (unsigned long long)sum, pos[0], pos[1], dir[0]);
    /* the observation gather through the ABI alone (RCCL inside the library), world of one rank:
     * what julia/BatchedSingleRoom.jl's gather_observations does */
    {
        unsigned char uid[RCW_UNIQUE_ID_BYTES];
        int32_t rank = -1, world = -1;
        void* all = NULL;
        uint32_t* back = (uint32_t*)malloc(npix * sizeof(uint32_t));
        CHECK(rcw_comm_unique_id(uid));
        CHECK(rcw_comm_init(h, uid, 0, 1));
        CHECK(rcw_comm_info(h, &rank, &world));
        if (rank != 0 || world != 1) { fprintf(stderr, "rcw_comm_info: %d/%d\n", rank, world); return 1; }
        CHECK(rcw_device_malloc(h, npix * sizeof(uint32_t), &all));
        for (int mode = RCW_GATHER_COLUMNS; mode <= RCW_GATHER_FRAMES; ++mode) {
            memset(back, 0, npix * sizeof(uint32_t));
            CHECK(rcw_gather_observations(h, mode, all));
            CHECK(rcw_memcpy_to_host(h, back, all, npix * sizeof(uint32_t)));
            if (memcmp(back, frames, npix * sizeof(uint32_t)) != 0) { fprintf(stderr, "gather mode %d differs from camera_view\n", mode); return 1; }
        }
        CHECK(rcw_device_free(h, all));
        CHECK(rcw_comm_destroy(h));
        free(back);
        printf("gather=ok\n");
    }
    free(frames);
    CHECK(rcw_destroy(h));
    /* the opt-in top view (update_top_view! SR:446-483) through the ABI alone: a second handle, the same action stream */
    {
        cfg.render_top_view = 1; cfg.pu_per_tu = 32;
        rcw_handle* ht = NULL;
        CHECK(rcw_create(&cfg, B, 0, 2024, &ht));
        int32_t form = -1;
        CHECK(rcw_set_top_view_form(ht, RCW_TOP_VIEW_TWO_KERNELS, 0));   /* (64 agents: the two-kernel form only when asked for) */
        CHECK(rcw_top_view_form(ht, &form));
        int32_t form_alone = -1;
        CHECK(rcw_update_top_view_form(ht, &form_alone));
        printf("top_view_form_alone=%d\n", (int)form_alone);
        char kname[64];
        CHECK(rcw_fill_kernel_name(ht, kname, (int32_t)sizeof kname));
        printf("fill_kernel=%s\n", kname);
        seed = 99;
        for (int s = 0; s < STEPS; ++s) {
            for (int a = 0; a < B; ++a) actions[a] = (uint8_t)(1 + lcg(&seed) % 4);
            CHECK(rcw_step(ht, actions));
        }
        const size_t tpix = (size_t)B * 256 * 256;
        uint32_t* top = (uint32_t*)malloc(tpix * sizeof(uint32_t));
        CHECK(rcw_top_view_copy(ht, top, 0, B));
        uint64_t tsum = 0;                                 /* position-weighted sum mod 2^64 (vectorisable on the checking side) */
        for (size_t k = 0; k < tpix; ++k) tsum += (uint64_t)top[k] * (2 * (uint64_t)k + 1);
        printf("top_view_form=%d top_checksum=%016llx\n", (int)form, (unsigned long long)tsum);
        free(top);
        CHECK(rcw_destroy(ht));
    }
    return 0;
}
