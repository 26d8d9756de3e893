/*
 * c_api_demo.c -- the C ABI of include/craftingworld.h used from plain C: no Python, no torch.
 *
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include -o examples/c_api_demo examples/c_api_demo.c \
 *       -L gym_craftingworld_amd -lcraftingworld -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../gym_craftingworld_amd' -Wl,-rpath,/opt/rocm/lib
 *   ./examples/c_api_demo
 *
 * Creates 1024 envs (21x21, dirty-cell pixel mode), seeds them like numpy RandomState(i), runs 600
 * random steps with auto-reset, reads back rewards/counters and one frame, and prints a checksum that
 * tests/test_c_api_demo.py compares with the CPU oracle's.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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
    return 0;
}
