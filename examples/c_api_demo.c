/*
 * c_api_demo.c -- the C ABI of include/craftingworld.h used from plain C: no Python, no torch.
 *
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include -o examples/c_api_demo examples/c_api_demo.c \
 *       -L gym_craftingworld_amd -lcraftingworld -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../gym_craftingworld_amd' -Wl,-rpath,/opt/rocm/lib
 *   ./examples/c_api_demo
 *
 * Creates 1024 envs (21x21, dirty-cell pixel mode), seeds them like numpy RandomState(i), runs 600
 * random steps with auto-reset, reads back rewards/counters and one frame, and prints a checksum that
 * tests/test_c_api_demo.py compares with the CPU oracle's.  Then the single-env loop of the reference
 * (docs/source/envs/gen_info.rst:62-82) through cw_step_resident: one env with host-mapped outputs, 3000
 * steps without a kernel launch (a resident kernel polls a doorbell word), reset() whenever done -- a
 * second line of checksums, and the time per step.
 */
#define _POSIX_C_SOURCE 199309L   /* clock_gettime under -std=c99 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "craftingworld.h"

#define CHECK_CW(x) do { int rc_ = (x); if (rc_ != CW_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, cw_last_error()); return 1; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

/* same tiny LCG on both sides of the test */
static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void)
{
    enum { N = 1024, T = 600, S = 21 };
    cw_task_menu menu = {0};
    menu.n_selected = 9; menu.number_of_tasks = 9; menu.stacking = 1; menu.reward_subset = 0;
    for (int i = 0; i < 9; i++) menu.selected_bits[i] = i;
    cw_config cfg = {0};
    cfg.abi_version = CW_ABI_VERSION; cfg.num_envs = N; cfg.size = S; cfg.max_steps = 300; cfg.n_task_list = 9;
    cfg.obs_mode = CW_OBS_PIXELS_DIRTY; cfg.auto_reset = 1; cfg.n_menus = 1; cfg.menus = &menu;
    cw_engine *eng = NULL;
    CHECK_CW(cw_create(&cfg, 0, &eng));

    uint32_t *seeds = (uint32_t *)malloc(N * sizeof(uint32_t));
    for (int i = 0; i < N; i++) seeds[i] = 5000u + (uint32_t)i;
    CHECK_CW(cw_seed_int(eng, seeds));

    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_CW(cw_reset(eng, st));

    uint8_t *h_act = (uint8_t *)malloc((size_t)T * N), *d_act = NULL;
    uint32_t rs = 12345u;
    for (size_t i = 0; i < (size_t)T * N; i++) h_act[i] = (uint8_t)(lcg(&rs) % 6u);
    CHECK_HIP(hipMalloc((void **)&d_act, (size_t)T * N));
    CHECK_HIP(hipMemcpy(d_act, h_act, (size_t)T * N, hipMemcpyHostToDevice));

    cw_buffer_table buf;
    CHECK_CW(cw_buffers(eng, &buf));
    int32_t *h_rew = (int32_t *)malloc(N * sizeof(int32_t));
    long long reward_sum = 0;
    for (int t = 0; t < T; t++) {
        CHECK_CW(cw_step(eng, d_act + (size_t)t * N, CW_ACT_U8, st));
        if (t % 100 == 99 || t == T - 1) {      /* stream-ordered read-back of the engine-owned buffer */
            CHECK_HIP(hipMemcpyAsync(h_rew, buf.reward, N * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            CHECK_HIP(hipStreamSynchronize(st));
            for (int i = 0; i < N; i++) reward_sum += h_rew[i];
        }
    }
    uint64_t counters[4];
    CHECK_HIP(hipMemcpy(counters, buf.counters, sizeof(counters), hipMemcpyDeviceToHost));
    uint8_t *frame = (uint8_t *)malloc(buf.frame_bytes);
    CHECK_HIP(hipMemcpy(frame, buf.obs + (size_t)17 * buf.frame_bytes, buf.frame_bytes, hipMemcpyDeviceToHost));
    uint32_t fnv = 2166136261u;
    for (size_t i = 0; i < buf.frame_bytes; i++) fnv = (fnv ^ frame[i]) * 16777619u;

    printf("envs %d steps %d env_steps %llu episodes %llu successes %llu reward_sum_sampled %lld frame17_fnv %08x\n", N, T,
           (unsigned long long)counters[0], (unsigned long long)counters[1], (unsigned long long)counters[2], reward_sum, fnv);
    CHECK_CW(cw_destroy(eng));

    /* ---- the single-env loop: step() = a doorbell word and a spin, no launch (cw_step_resident) ---- */
    enum { T1 = 3000 };
    cfg.num_envs = 1; cfg.max_steps = 120; cfg.auto_reset = 0; cfg.host_outputs = 1;     /* gym.Env semantics; outputs in pinned host memory */
    cw_engine *one = NULL;
    CHECK_CW(cw_create(&cfg, 0, &one));
    uint32_t seed1 = 7777u;
    CHECK_CW(cw_seed_int(one, &seed1));
    CHECK_CW(cw_reset(one, st));
    CHECK_CW(cw_synchronize(one, st));
    cw_buffer_table hb;
    CHECK_CW(cw_buffers(one, &hb));                      /* host pointers: reward, done, obs are readable as plain memory */
    long long rsum = 0;
    int episodes = 0;
    uint32_t rs1 = 99u, trace = 2166136261u;
    struct timespec ta, tb;
    clock_gettime(CLOCK_MONOTONIC, &ta);
    for (int t = 0; t < T1; t++) {
        CHECK_CW(cw_step_resident(one, (int32_t)(lcg(&rs1) % 6u), 0));
        rsum += hb.reward[0];
        trace = (trace ^ (uint32_t)hb.achieved[0]) * 16777619u;
        if (hb.done[0]) {                                /* the caller resets, as in the reference loop (this parks the resident kernel) */
            episodes++;
            CHECK_CW(cw_reset(one, st));
            CHECK_CW(cw_synchronize(one, st));
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &tb);
    uint32_t fnv1 = 2166136261u;
    for (size_t i = 0; i < hb.frame_bytes; i++) fnv1 = (fnv1 ^ hb.obs[i]) * 16777619u;
    printf("single_env steps %d episodes %d reward_sum %lld achieved_trace %08x frame_fnv %08x us_per_step_incl_resets %.2f\n", T1, episodes, rsum, trace,
           fnv1, ((tb.tv_sec - ta.tv_sec) * 1e9 + (tb.tv_nsec - ta.tv_nsec)) / 1e3 / T1);
    CHECK_CW(cw_destroy(one));
    return 0;
}
