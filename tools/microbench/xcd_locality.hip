// Micro-benchmark (GPU box only): does it matter WHICH XCD writes which address?  Workgroups go round-robin to the 8 XCDs
// (block b -> XCD b % 8).  One 16-B store per thread, 256 threads = one 4-KiB chunk per block (torch's fill_ shape);
// the chunk a block writes is permuted so that XCD k covers a different address residue class.
//   hipcc -O3 --offload-arch=gfx950 -o xcd_locality xcd_locality.hip && ./xcd_locality
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// mode 0: chunk = b ^ x (permute within groups of 8)   mode 1: chunk = (b % 8) * (nb / 8) + b / 8 (XCD k owns one contiguous eighth)
// mode 2: chunk = bit-rotated so that XCD k owns residues of a coarser granule (shift = log2 granule in chunks)
__global__ __launch_bounds__(256) void fill1(u32x4 *dst, uint32_t nb, int mode, uint32_t x, uint32_t shift)
{
    uint32_t b = blockIdx.x, c;
    if (mode == 0) c = b ^ x;
    else if (mode == 1) c = (b & 7u) * (nb >> 3) + (b >> 3);
    else {   // XCD id (low 3 bits of b) is moved up to bits [shift, shift+3) of the chunk index
        const uint32_t xcd = b & 7u, rest = b >> 3;
        const uint32_t lo = rest & ((1u << shift) - 1u), hi = rest >> shift;
        c = (hi << (shift + 3)) | (xcd << shift) | lo;
    }
    if (c >= nb) return;
    const u32x4 d = {b, 1, 2, 3};
    dst[(size_t)c * 256 + threadIdx.x] = d;
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

int main()
{
    const uint32_t nb = 1u << 18;                      // 2^18 chunks x 4 KiB = 1 GiB
    const size_t bytes = (size_t)nb * 4096;
    u32x4 *buf; CHECK(hipMalloc(&buf, bytes));
    for (int rep = 0; rep < 2; rep++) {
        for (uint32_t x = 0; x < 8; x++) {
            const float t = bench([&] { hipLaunchKernelGGL(fill1, dim3(nb), dim3(256), 0, 0, buf, nb, 0, x, 0u); });
            printf("chunk = b ^ %u                         %.3f ms %5.0f GB/s\n", x, t, bytes / t / 1e6);
        }
        { const float t = bench([&] { hipLaunchKernelGGL(fill1, dim3(nb), dim3(256), 0, 0, buf, nb, 1, 0u, 0u); });
          printf("XCD k owns contiguous eighth k         %.3f ms %5.0f GB/s\n", t, bytes / t / 1e6); }
        for (uint32_t sh = 0; sh <= 10; sh++) {
            const float t = bench([&] { hipLaunchKernelGGL(fill1, dim3(nb), dim3(256), 0, 0, buf, nb, 2, 0u, sh); });
            printf("XCD k owns residue k of %5u-KiB granules  %.3f ms %5.0f GB/s\n", 4u << sh, t, bytes / t / 1e6);
        }
    }
    return 0;
}
