// GPU box only: does gfx950 execute scalar-memory atomics (s_atomic_add, lgkmcnt domain)?  Each wave takes tickets from one
// counter with s_atomic_add ... glc; all tickets must be distinct and cover 0..total-1.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
__global__ void k(unsigned *ctr, unsigned *out, int per_wave)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < per_wave; i++) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if ((threadIdx.x & 63) == 0) out[wave * per_wave + i] = v;
    }
}
int main()
{
    const int waves = 1024, per = 16, total = waves * per;
    unsigned *ctr, *out;
    CHECK(hipMalloc(&ctr, 4)); CHECK(hipMalloc(&out, total * 4));
    CHECK(hipMemset(ctr, 0, 4)); CHECK(hipMemset(out, 0xff, total * 4));
    hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, ctr, out, per);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> h(total); unsigned c = 0;
    CHECK(hipMemcpy(h.data(), out, total * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    bool ok = (c == (unsigned)total);
    for (int i = 0; i < total; i++) ok = ok && h[i] == (unsigned)i;
    printf("counter %u (expected %d), tickets %s\n", c, total, ok ? "distinct and complete: scalar atomics work" : "WRONG");
    return ok ? 0 : 2;
}
