/* examples/host.c — a host in plain C over the C ABI (include/soccer_hip.h): no Python, no torch, no HIP headers.
 *
 * 65 536 environments of the reference's default game (5x4 pitch, SoccerSimultaneousEnv(width=5, height=4, slip_prob),
 * gym_soccer/envs/soccer_simultaneous_env.py:35), reset, then T steps with both players acting uniformly at random —
 * drawn in the kernel from the lanes' Philox words — in ONE fused launch, the [T][n] trajectories reduced on the device to
 * per-lane episode returns and the (-1, 0, +1) histogram.  Prints one line; tests/test_gpu_c_host.py builds this with gcc,
 * runs it and checks the line against the same run through the Python layer (same seed => same counts, to the unit).
 *
 * Build:  gcc -O2 -Iinclude examples/host.c -o build/host_c -Lgym_soccer_littman94_amd -lsoccer_hip -Wl,-rpath,$PWD/gym_soccer_littman94_amd
 * Run:    build/host_c [slip_prob [seed [T]]]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "soccer_hip.h"

#define CHECK(h, call) do { int rc_ = (call); if (rc_ != SOCCER_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, soccer_last_error(h)); return 1; } } while (0)

int main(int argc, char** argv) {
    const double slip = argc > 1 ? atof(argv[1]) : 0.2;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], NULL, 10) : 7;
    const int T = argc > 3 ? atoi(argv[3]) : 100;
    const uint64_t n = 65536;

    soccer_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_lanes = n; cfg.width = 5; cfg.height = 4; cfg.slip_prob = slip; cfg.max_steps = 100;
    cfg.device = 0; cfg.seed = seed; cfg.lane_offset = 0; cfg.flags = SOCCER_F_AUTORESET;
    soccer_handle* h = NULL;
    CHECK(NULL, soccer_create(&cfg, &h));

    void *rew = NULL, *term = NULL, *trunc = NULL, *last = NULL, *count = NULL;
    CHECK(h, soccer_malloc(h, (size_t)T * n, &rew));
    CHECK(h, soccer_malloc(h, (size_t)T * n, &term));
    CHECK(h, soccer_malloc(h, (size_t)T * n, &trunc));
    CHECK(h, soccer_malloc(h, n, &last));
    CHECK(h, soccer_malloc(h, n * sizeof(int32_t), &count));

    CHECK(h, batched_reset(h, NULL, NULL, NULL));                       /* reset(): every lane draws its start state (:410-424) */
    soccer_rollout_args ro;
    memset(&ro, 0, sizeof ro);
    ro.n_steps = T; ro.sample_actions = 1;                               /* step() x T (:375-408), actions sampled in the kernel */
    ro.reward = (int8_t*)rew; ro.terminated = (uint8_t*)term; ro.truncated = (uint8_t*)trunc; ro.out_stride = (int64_t)n;
    CHECK(h, batched_rollout(h, &ro));
    uint64_t hist[3] = {0, 0, 0};
    CHECK(h, soccer_trajectory_returns(h, T, (const int8_t*)rew, (const uint8_t*)term, (const uint8_t*)trunc, (int64_t)n,
                                       (int8_t*)last, (int32_t*)count, hist));   /* synchronises: hist is on the host */

    int32_t* counts = (int32_t*)malloc(n * sizeof(int32_t));
    int8_t* lasts = (int8_t*)malloc(n);
    CHECK(h, soccer_memcpy_d2h(h, counts, count, n * sizeof(int32_t)));
    CHECK(h, soccer_memcpy_d2h(h, lasts, last, n));
    uint64_t episodes = 0; long long last_sum = 0;
    for (uint64_t i = 0; i < n; ++i) { episodes += (uint64_t)counts[i]; last_sum += lasts[i]; }
    uint64_t lib_hist[3], misuse = 0;
    CHECK(h, soccer_get_stats(h, lib_hist, &misuse));                    /* the rollout's own episode histogram */

    printf("lanes %llu steps %d slip %g seed %llu tick %llu episodes %llu hist %llu %llu %llu last_sum %lld misuse %llu\n",
           (unsigned long long)n, T, slip, (unsigned long long)seed, (unsigned long long)soccer_tick(h),
           (unsigned long long)episodes, (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2],
           last_sum, (unsigned long long)misuse);
    const int ok = episodes == hist[0] + hist[1] + hist[2] && lib_hist[0] == hist[0] && lib_hist[1] == hist[1] && lib_hist[2] == hist[2] && misuse == 0;
    free(counts); free(lasts);
    soccer_free(h, rew); soccer_free(h, term); soccer_free(h, trunc); soccer_free(h, last); soccer_free(h, count);
    soccer_destroy(h);
    return ok ? 0 : 2;
}
