// Micro-benchmark (GPU box only): what limits the frame-render kernel's HBM write rate?
// Fills a 65536 x 21168 B buffer with several store shapes / ALU loads and prints GB/s each.
//   hipcc -O3 --offload-arch=gfx950 -o store_variants store_variants.hip && ./store_variants
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// A: pure fill, 16 B per lane, grid-stride
template <int NT>
__global__ __launch_bounds__(256) void fill_x4(u32x4 *dst, size_t n16, uint32_t v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 d = {v, v + 1, v + 2, v + 3};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(d, dst + i); else dst[i] = d;
    }
}
// B: pure fill, 12 B per lane
template <int NT>
__global__ __launch_bounds__(256) void fill_x3(uint8_t *dst, size_t n12, uint32_t v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x3 d = {v, v + 1, v + 2};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n12; i += stride) {
        if (NT) __builtin_nontemporal_store(d, (u32x3_a4 *)(dst + i * 12)); else *(u32x3_a4 *)(dst + i * 12) = d;
    }
}
// C: wave-per-frame like cw_render_kernel: contiguous 768 B per wave store, frame by frame, x3
template <int NT, int ALU>
__global__ __launch_bounds__(256) void frame_x3(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_items = frame_bytes / 12;
    for (int f = wave; f < n_frames; f += n_waves) {
        uint8_t *base = dst + (size_t)f * frame_bytes;
        uint4 p = pos[f];
        uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        for (uint32_t it = lane; it < n_items; it += 64) {
            uint32_t col = 0;
            if (ALU) {
                const uint32_t y = __umulhi(it, 204522253u);   // /21
                const uint32_t c = it - y * 21;
                const uint32_t cell = (y >> 2) * 21 + c;
#pragma unroll
                for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
            } else col = it;
            u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
            if (NT) __builtin_nontemporal_store(d, (u32x3_a4 *)(base + (size_t)it * 12)); else *(u32x3_a4 *)(base + (size_t)it * 12) = d;
        }
    }
}
// D: wave-per-frame, 16 B per lane, trivial ALU
template <int NT>
__global__ __launch_bounds__(256) void frame_x4(uint8_t *dst, int n_frames, uint32_t frame_bytes)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_items = frame_bytes / 16;
    for (int f = wave; f < n_frames; f += n_waves) {
        uint8_t *base = dst + (size_t)f * frame_bytes;
        for (uint32_t it = lane; it < n_items; it += 64) {
            u32x4 d = {it, it + 1, it + 2, it + 3};
            if (NT) __builtin_nontemporal_store(d, (u32x4 *)(base + (size_t)it * 16)); else *(u32x4 *)(base + (size_t)it * 16) = d;
        }
    }
}


// E: block-per-frame (256 threads stride 256 items): neighbouring 768-B chunks are written by sibling waves
template <int ALU>
__global__ __launch_bounds__(256) void frame_block_x3(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const uint32_t n_items = frame_bytes / 12;
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        uint8_t *base = dst + (size_t)f * frame_bytes;
        uint4 p = pos[f];
        uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        for (uint32_t it = threadIdx.x; it < n_items; it += 256) {
            uint32_t col = it;
            if (ALU) {
                const uint32_t y = __umulhi(it, 204522253u);
                const uint32_t c = it - y * 21;
                const uint32_t cell = (y >> 2) * 21 + c;
                col = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
            }
            u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
            *(u32x3_a4 *)(base + (size_t)it * 12) = d;
        }
    }
}
// F: one flat array of N*1764 items, grid-stride, per-lane env lookup (frames are packed back to back)
__global__ __launch_bounds__(256) void flat_x3(uint8_t *dst, int n_frames, const uint4 *pos)
{
    const size_t total = (size_t)n_frames * 1764;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const uint32_t f = (uint32_t)(g / 1764);
        const uint32_t it = (uint32_t)(g - (size_t)f * 1764);
        uint4 p = pos[f];
        uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        const uint32_t y = __umulhi(it, 204522253u);
        const uint32_t c = it - y * 21;
        const uint32_t cell = (y >> 2) * 21 + c;
        uint32_t col = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
        u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
        *(u32x3_a4 *)(dst + g * 12) = d;
    }
}
// G: cell-row form: lane = (row-in-group rr 0..2, col c 0..20), colour computed once, 4 pixel-row stores
__global__ __launch_bounds__(256) void frame_cellrow_x3(uint8_t *dst, int n_frames, uint32_t frame_bytes, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const int rr = lane / 21, c = lane - rr * 21;
    for (int f = wave; f < n_frames; f += n_waves) {
        uint8_t *base = dst + (size_t)f * frame_bytes;
        uint4 p = pos[f];
        uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        if (lane < 63)
            for (int r0 = 0; r0 < 21; r0 += 3) {
                const uint32_t r = r0 + rr;
                const uint32_t cell = r * 21 + c;
                uint32_t col = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
                u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
                uint8_t *q = base + (size_t)(4 * r) * 252 + 12 * c;
#pragma unroll
                for (int dy = 0; dy < 4; dy++) *(u32x3_a4 *)(q + dy * 252) = d;
            }
    }
}

// G2: like G but pixel rows padded to 256 B (frame pitch 84*256 = 21504 B = 168 whole lines) and the
// 4 pad bytes written too, so every 128-B line is written whole by one store instruction
__global__ __launch_bounds__(256) void frame_cellrow_padded(uint8_t *dst, int n_frames, const uint4 *pos)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const int rr = lane / 21, c = lane - rr * 21;
    for (int f = wave; f < n_frames; f += n_waves) {
        uint8_t *base = dst + (size_t)f * 21504;
        uint4 p = pos[f];
        uint32_t sp[8] = {p.x & 0xffff, p.x >> 16, p.y & 0xffff, p.y >> 16, p.z & 0xffff, p.z >> 16, p.w & 0xffff, p.w >> 16};
        if (lane < 63)
            for (int r0 = 0; r0 < 21; r0 += 3) {
                const uint32_t r = r0 + rr;
                const uint32_t cell = r * 21 + c;
                uint32_t col = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) col = (cell == sp[k]) ? (0x112233u * (k + 1)) : col;
                u32x3 d = {col | (col << 24), (col >> 8) | (col << 16), (col >> 16) | (col << 8)};
                uint8_t *q = base + (size_t)(4 * r) * 256 + 12 * c;
#pragma unroll
                for (int dy = 0; dy < 4; dy++) {
                    *(u32x3_a4 *)(q + dy * 256) = d;
                    if (c == 20) *(uint32_t *)(q + dy * 256 + 12) = 0;
                }
            }
    }
}

template <typename F>
static void bench(const char *name, size_t bytes, F launch)
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
    printf("%-44s median %.3f ms  %.0f GB/s   (min %.3f ms %.0f GB/s)\n", name, ms[7], bytes / ms[7] / 1e6, ms[0], bytes / ms[0] / 1e6);
}

int main(int argc, char **argv)
{
    const int alloc_mode = argc > 1 ? atoi(argv[1]) : 0;   // 0 hipMalloc, 1 uncached, 2 fine-grained
    const int N = 65536; const uint32_t FB = 21168;
    const size_t bytes = (size_t)N * FB;
    uint8_t *buf; uint4 *pos;
    uint8_t *buf2; CHECK(hipMalloc(&buf2, (size_t)N * 21504));
    if (alloc_mode == 1) CHECK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocUncached));
    else if (alloc_mode == 2) CHECK(hipExtMallocWithFlags((void **)&buf, bytes, hipDeviceMallocFinegrained));
    else CHECK(hipMalloc(&buf, bytes));
    printf("alloc mode %d\n", alloc_mode);
    CHECK(hipMalloc(&pos, N * 16)); CHECK(hipMemset(pos, 7, N * 16));
    for (int blocks : {256, 1024}) {
        printf("-- grid %d blocks x 256\n", blocks);
        bench("fill x4 plain", bytes, [&] { hipLaunchKernelGGL(fill_x4<0>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16, 1u); });
        bench("fill x4 nontemporal", bytes, [&] { hipLaunchKernelGGL(fill_x4<1>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16, 1u); });
        bench("fill x3 plain", bytes, [&] { hipLaunchKernelGGL(fill_x3<0>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 12, 1u); });
        bench("fill x3 nontemporal", bytes, [&] { hipLaunchKernelGGL(fill_x3<1>, dim3(blocks), dim3(256), 0, 0, buf, bytes / 12, 1u); });
        bench("frame x3 plain, no ALU", bytes, [&] { hipLaunchKernelGGL((frame_x3<0, 0>), dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("frame x3 nt, no ALU", bytes, [&] { hipLaunchKernelGGL((frame_x3<1, 0>), dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("frame x3 plain, render ALU", bytes, [&] { hipLaunchKernelGGL((frame_x3<0, 1>), dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("frame x3 nt, render ALU", bytes, [&] { hipLaunchKernelGGL((frame_x3<1, 1>), dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("E block-per-frame x3, no ALU", bytes, [&] { hipLaunchKernelGGL(frame_block_x3<0>, dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("E block-per-frame x3, render ALU", bytes, [&] { hipLaunchKernelGGL(frame_block_x3<1>, dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("F flat items x3, render ALU", bytes, [&] { hipLaunchKernelGGL(flat_x3, dim3(blocks), dim3(256), 0, 0, buf, N, pos); });
        bench("G cell-row x3 (4 stores/cell)", bytes, [&] { hipLaunchKernelGGL(frame_cellrow_x3, dim3(blocks), dim3(256), 0, 0, buf, N, FB, pos); });
        bench("G2 cell-row, rows padded to 256 B", (size_t)N * 21504, [&] { hipLaunchKernelGGL(frame_cellrow_padded, dim3(blocks), dim3(256), 0, 0, buf2, N, pos); });
        bench("frame x4 plain", bytes, [&] { hipLaunchKernelGGL(frame_x4<0>, dim3(blocks), dim3(256), 0, 0, buf, N, FB); });
        bench("frame x4 nt", bytes, [&] { hipLaunchKernelGGL(frame_x4<1>, dim3(blocks), dim3(256), 0, 0, buf, N, FB); });
    }
    return 0;
}
