// Micro-benchmark (GPU box only): the frame paint's store pattern (one wave per 21x21 frame, lane = cell, 4 x 12-B stores per
// 64-cell group, 7 groups per frame) with the ORDER of the groups staggered between waves, so that concurrently running
// waves are not all at the same relative offset of their frames.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// rot_mode: 0 none | 1 wave % 7 | 2 (wave / 4) % 7 | 3 frame % 7 | 4 reversed groups | 5 (wave * 3) % 7 | 6 hash
template <int RM>
__global__ __launch_bounds__(256) void frame_cells(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int f = wave; f < n_frames; f += n_waves) {
        uint8_t *base = dst + (size_t)f * frame_bytes;
        const uint4 p = pos[f & 1023];
        const uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        uint32_t rot = 0;
        if (RM == 1) rot = wave % 7;
        if (RM == 2) rot = (wave / 4) % 7;
        if (RM == 3) rot = f % 7;
        if (RM == 5) rot = (wave * 3) % 7;
        if (RM == 6) rot = ((uint32_t)wave * 2654435761u >> 16) % 7;
        for (uint32_t g0 = 0; g0 < 7; g0++) {
            uint32_t g = g0 + rot; g = g >= 7 ? g - 7 : g;
            if (RM == 4) g = 6 - g0;
            const uint32_t cell = g * 64 + lane;
            if (cell < 441) {
                const uint32_t r = __umulhi(cell, 204522253u), c = cell - r * 21;
                uint32_t col = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
                const u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
                uint8_t *q = base + (size_t)(4 * r) * 252 + 12 * c;
#pragma unroll
                for (int dy = 0; dy < 4; dy++) *(u32x3_a4 *)(q + dy * 252) = d;
            }
        }
    }
}
// row-aligned variant: 63 lanes = 3 whole grid rows per group (7 groups), so every store instruction writes 3 complete
// 252-byte pixel-row segments and no segment is split between two groups
__global__ __launch_bounds__(256) void frame_rows(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int f = wave; f < n_frames; f += n_waves) {
        uint8_t *base = dst + (size_t)f * frame_bytes;
        const uint4 p = pos[f & 1023];
        const uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        for (uint32_t g = 0; g < 7; g++) {
            const uint32_t cell = g * 63 + lane;
            if (lane < 63) {
                const uint32_t r = __umulhi(cell, 204522253u), c = cell - r * 21;
                uint32_t col = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
                const u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
                uint8_t *q = base + (size_t)(4 * r) * 252 + 12 * c;
#pragma unroll
                for (int dy = 0; dy < 4; dy++) *(u32x3_a4 *)(q + dy * 252) = d;
            }
        }
    }
}

// torch-fill shape: NOT persistent; a 256-thread block = one 63-cell group (3 grid rows) of one frame, wave dy of the block
// writes pixel row dy of those cells: ONE 12-byte store per lane, the block's 4 waves cover 3024 contiguous bytes.
// Records by scalar loads; colours from a 9-entry table held in lanes of a VGPR (v_readlane by code).
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void frame_torchlike(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const uint32_t dy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t f = blockIdx.x / 7u, g = blockIdx.x - f * 7u;
    const u32x4s p = *(const __attribute__((address_space(4))) u32x4s *)(pos + (f & 1023u));
    const uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
    if (lane < 63) {
        const uint32_t cell = g * 63 + lane;
        const uint32_t r = __umulhi(cell, 204522253u), c = cell - r * 21;
        uint32_t col = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
        const u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
        uint8_t *q = dst + (size_t)f * frame_bytes + (size_t)(4 * r + dy) * 252 + 12 * c;
        *(u32x3_a4 *)q = d;
    }
}
// the same with 2 or 4 frames' worth of work per block kept torch-like: block = (frame, group), wave = pixel row, 64-thread blocks
__global__ __launch_bounds__(64) void frame_torchlike64(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x;                       // one wave per (frame, group, dy)
    const uint32_t fg = w >> 2, dy = w & 3u;
    const uint32_t f = fg / 7u, g = fg - f * 7u;
    const u32x4s p = *(const __attribute__((address_space(4))) u32x4s *)(pos + (f & 1023u));
    const uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
    if (lane < 63) {
        const uint32_t cell = g * 63 + lane;
        const uint32_t r = __umulhi(cell, 204522253u), c = cell - r * 21;
        uint32_t col = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
        const u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
        uint8_t *q = dst + (size_t)f * frame_bytes + (size_t)(4 * r + dy) * 252 + 12 * c;
        *(u32x3_a4 *)q = d;
    }
}

template <typename F>
static float bench(F launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    std::vector<float> ms;
    for (int i = 0; i < 15; i++) {
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float t; CHECK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[7];
}
template <int RM>
static void run(const char *name, uint8_t *buf, const uint4 *pos, int N, uint32_t FB)
{
    const float t = bench([&] { hipLaunchKernelGGL(frame_cells<RM>, dim3(256), dim3(256), 0, 0, buf, N, FB, pos); });
    printf("%-28s %.3f ms %5.0f GB/s\n", name, t, (size_t)N * FB / t / 1e6);
}
int main()
{
    const int N = 65536; const uint32_t FB = 21168;
    uint8_t *buf; uint4 *pos;
    CHECK(hipMalloc(&buf, (size_t)N * FB));
    CHECK(hipMalloc(&pos, 1024 * 16)); CHECK(hipMemset(pos, 7, 1024 * 16));
    for (int rep = 0; rep < 3; rep++) {
        { const float t = bench([&] { hipLaunchKernelGGL(frame_torchlike, dim3(N * 7), dim3(256), 0, 0, buf, N, FB, pos); });
          printf("%-28s %.3f ms %5.0f GB/s\n", "torch-like, 256-thr blocks", t, (size_t)N * FB / t / 1e6); }
        { const float t = bench([&] { hipLaunchKernelGGL(frame_torchlike64, dim3(N * 28), dim3(64), 0, 0, buf, N, FB, pos); });
          printf("%-28s %.3f ms %5.0f GB/s\n", "torch-like, 64-thr blocks", t, (size_t)N * FB / t / 1e6); }
        run<0>("in order", buf, pos, N, FB);
        { const float t = bench([&] { hipLaunchKernelGGL(frame_rows, dim3(256), dim3(256), 0, 0, buf, N, FB, pos); });
          printf("%-28s %.3f ms %5.0f GB/s\n", "63 lanes = 3 whole rows", t, (size_t)N * FB / t / 1e6); }
        run<1>("start at wave % 7", buf, pos, N, FB);
        run<2>("start at (wave/4) % 7", buf, pos, N, FB);
        run<3>("start at frame % 7", buf, pos, N, FB);
        run<4>("reversed", buf, pos, N, FB);
        run<5>("start at (3 wave) % 7", buf, pos, N, FB);
        run<6>("start at hash(wave) % 7", buf, pos, N, FB);
    }
    return 0;
}
