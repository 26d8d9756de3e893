// Micro-benchmark (GPU box only): the linear sweep's STORE SHAPE with no loads and no ALU, each shape at every pacing -- what is the
// ceiling of each shape once the rate is right?  Same launch geometry as cw_render_kernel (persistent, 256 workgroups x 4 waves,
// jobs round-robin in address order over a 65536 x 21168 B buffer).
//   shape 0: 63 lanes, 4 stores of 12 B at the cell's 4 pixel rows (3 runs of 252 B per instruction)      [the product]
//   shape 1: 63 lanes, 4 stores of 12 B, each 756 B contiguous                                              [CW_TUNE_RENDER_SHAPE=3]
//   shape 2: 63 lanes, 3 stores of 16 B, each 1008 B contiguous                                             [CW_TUNE_RENDER_SHAPE=4]
//   shape 3: 64 lanes, 4 stores of 12 B, 768 B contiguous and 256-B aligned (jobs of 3072 B over the flat byte array)
//   shape 4: 64 lanes, 3 stores of 16 B, 1024 B contiguous and aligned (jobs of 3072 B)
//   shape 5: 64 lanes, 1 store of 12 B per job (jobs of 768 B): a plain fill handed out the same way
// and WHAT is written (zeros, one constant, per-lane values, per-lane per-job values).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256) void sweep(uint8_t *dst, size_t bytes, int pace_pair, int pace_mid, int data)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    constexpr uint32_t JOB = SHAPE <= 2 ? 3024u : SHAPE == 5 ? 768u : 3072u;
    dst += (data >> 8);                 // (alignment experiment: every chunk shifted by this many bytes)
    data &= 0xFF;
    const uint32_t n_jobs = (uint32_t)(bytes / JOB);
    const uint32_t rr = lane / 21, c = lane - rr * 21;
    // what is written: 0 zeros | 1 one constant everywhere | 2 a different value per lane | 3 a different value per lane and job
    u32x3 d3 = {0u, 0u, 0u};
    u32x4 d4 = {0u, 0u, 0u, 0u};
    if (data == 1) { d3 = u32x3{0x2745456Eu, 0x456E2745u, 0x6E274545u}; d4 = u32x4{0x2745456Eu, 0x456E2745u, 0x6E274545u, 0x2745456Eu}; }
    if (data >= 2) { const uint32_t h = (uint32_t)lane * 2654435761u; d3 = u32x3{h, h * 3u, h ^ 0x5bd1e995u}; d4 = u32x4{h, h * 3u, h ^ 0x5bd1e995u, h * 7u}; }
    int k = 0;
    for (uint32_t j = wave; j < n_jobs; j += n_waves, k++) {
        if (data == 3) { const uint32_t h = j * 0x9E3779B9u; d3.x ^= h; d3.y += h; d3.z ^= h >> 7; d4.x ^= h; d4.y += h; d4.z ^= h >> 7; d4.w += h; }
        if (k & 1) for (int z = 0; z < pace_pair; z++) __builtin_amdgcn_s_sleep(1);
        uint8_t *q = dst + (size_t)j * JOB;
        if (SHAPE == 0) {
            if (lane < 63) {
                uint8_t *p = q + rr * 1008u + c * 12u;
                *(u32x3_a4 *)(p) = d3;
                *(u32x3_a4 *)(p + 252) = d3;
                if (pace_mid) __builtin_amdgcn_s_sleep(1);
                *(u32x3_a4 *)(p + 504) = d3;
                *(u32x3_a4 *)(p + 756) = d3;
            }
        } else if (SHAPE == 1 || SHAPE == 3) {
            const uint32_t L = SHAPE == 1 ? 63u : 64u;
            if ((uint32_t)lane < L) {
                uint8_t *p = q + lane * 12u;
                *(u32x3_a4 *)(p) = d3;
                *(u32x3_a4 *)(p + 12u * L) = d3;
                if (pace_mid) __builtin_amdgcn_s_sleep(1);
                *(u32x3_a4 *)(p + 24u * L) = d3;
                *(u32x3_a4 *)(p + 36u * L) = d3;
            }
        } else if (SHAPE == 2 || SHAPE == 4) {
            const uint32_t L = SHAPE == 2 ? 63u : 64u;
            if ((uint32_t)lane < L) {
                uint8_t *p = q + lane * 16u;
                *(u32x4_a4 *)(p) = d4;
                *(u32x4_a4 *)(p + 16u * L) = d4;
                if (pace_mid) __builtin_amdgcn_s_sleep(1);
                *(u32x4_a4 *)(p + 32u * L) = d4;
            }
        } else if (SHAPE == 6) {            // shape 3 with the flat render's arithmetic: unit -> cell, 8 compares, overlay, per store
            const uint32_t sp0 = j * 7u, u0 = j * 256u + lane;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                uint32_t f = u0 + 64u * s4;
                f = f >= 1764u ? f - 1764u : f;
                const uint32_t y = __umulhi(f, 204522253u), cx = f - y * 21u, cell = (y >> 2) * 21u + cx;
                uint32_t col = 0;
#pragma unroll
                for (int o = 0; o < 8; o++) col = (cell == ((sp0 + 53u * o) & 511u)) ? (0x112233u * (o + 1)) : col;
                const uint32_t ov = (cell == (sp0 & 255u) && ((y & 3u) == 1u || (y & 3u) == 2u)) ? 0xFFFFFFu : col;
                u32x3 v = {__builtin_amdgcn_perm(ov, col, 0x04020100u) ^ d3.x, __builtin_amdgcn_perm(ov, col, 0x05040605u), __builtin_amdgcn_perm(ov, col, 0x02010006u)};
                *(u32x3_a4 *)(q + lane * 12u + 768u * s4) = v;
                if (s4 == 1 && pace_mid) __builtin_amdgcn_s_sleep(1);
            }
        } else {
            *(u32x3_a4 *)(q + lane * 12u) = d3;
            if (pace_mid && (k & 1)) __builtin_amdgcn_s_sleep(1);
        }
    }
}

static float time_launches(void (*kern)(uint8_t *, size_t, int, int, int), int blocks, uint8_t *buf, size_t bytes, int pair, int mid, int data, float *mn)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, bytes, pair, mid, data);
    std::vector<float> ms;
    for (int i = 0; i < 15; i++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, bytes, pair, mid, data);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float t; CHECK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    *mn = ms[0];
    return ms[7];
}

int main(int argc, char **argv)
{
    const int N = 65536; const uint32_t FB = 21168;
    const size_t bytes = (size_t)N * FB;
    uint8_t *buf;
    CHECK(hipMalloc(&buf, bytes + 4096));
    void (*kerns[7])(uint8_t *, size_t, int, int, int) = {sweep<0>, sweep<1>, sweep<2>, sweep<3>, sweep<4>, sweep<5>, sweep<6>};
    const char *names[7] = {"0 cell rows 4x(3x252 B)", "1 contiguous 4x756 B", "2 contiguous 3x1008 B", "3 aligned 4x768 B", "4 aligned 3x1024 B", "5 fill 1x768 B per job", "6 aligned 4x768 B + ALU"};
    const char *data_names[4] = {"zeros", "one constant", "per-lane values", "per-lane per-job values"};
    const int only_blocks = argc > 1 ? atoi(argv[1]) : 0;
    // alignment: the aligned shapes with every chunk shifted by 16 .. 256 bytes
    for (int off : {0, 16, 32, 64, 128, 256}) {
        float mn;
        printf("chunks shifted by %3d B:  4x768 B %.4f ms   3x1024 B %.4f ms   (1024 workgroups, unpaced, zeros)\n", off,
               time_launches(kerns[3], 1024, buf, bytes, 0, 0, off << 8, &mn), time_launches(kerns[4], 1024, buf, bytes, 0, 0, off << 8, &mn));
    }
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        float mn;
        printf("%4d workgroups: cell rows %.4f   aligned 4x768 %.4f   aligned 4x768 + ALU %.4f   aligned 3x1024 %.4f  (unpaced) | m+0: %.4f %.4f %.4f %.4f\n", blocks,
               time_launches(kerns[0], blocks, buf, bytes, 0, 0, 2, &mn), time_launches(kerns[3], blocks, buf, bytes, 0, 0, 2, &mn),
               time_launches(kerns[6], blocks, buf, bytes, 0, 0, 2, &mn), time_launches(kerns[4], blocks, buf, bytes, 0, 0, 2, &mn),
               time_launches(kerns[0], blocks, buf, bytes, 0, 1, 2, &mn), time_launches(kerns[3], blocks, buf, bytes, 0, 1, 2, &mn),
               time_launches(kerns[6], blocks, buf, bytes, 0, 1, 2, &mn), time_launches(kerns[4], blocks, buf, bytes, 0, 1, 2, &mn));
    }
    for (int data = 0; data < 4; data += 3)
    for (int blocks : {256, 1024}) {
        if (only_blocks && blocks != only_blocks) continue;
        printf("-- DATA %s; %d workgroups x 4 waves; median ms of 15 launches, by pacing [mid]+pair\n", data_names[data], blocks);
        for (int s = 0; s < 7; s++) {
            printf("%-26s", names[s]);
            float best = 1e9f; int bp = 0, bm = 0;
            for (int mid = 0; mid < 2; mid++)
                for (int pair : {0, 1, 2, 3, 4, 6, 8}) {
                    float mn;
                    const float t = time_launches(kerns[s], blocks, buf, bytes, pair, mid, data, &mn);
                    printf(" %s%d:%.3f", mid ? "m+" : "", pair, t);
                    if (t < best) { best = t; bp = pair; bm = mid; }
                }
            printf("   BEST %s%d %.4f ms = %.2f TB/s\n", bm ? "m+" : "", bp, best, bytes / best / 1e9);
        }
    }
    return 0;
}
