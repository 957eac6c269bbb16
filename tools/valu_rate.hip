// valu_rate.hip — micro-benchmark: issue rate of a few VALU ops on gfx950 (not part of the product).
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    const uint32_t m = seed | 0x10101u;
    for (int i = 0; i < iters; i++) {
#define STEP(x)                                                           \
    if (OP == 0) x = x + (x & m);                                         \
    else if (OP == 1) x = __popc(x & m) + x;                              \
    else if (OP == 2) x = __builtin_amdgcn_ubfe(x, 3, 7) + (x ^ m);       \
    else x = x * m + 1u;
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
static void run(const char* name, uint32_t* d) {
    const int iters = 4096, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 8 * blocks * threads / 64;  // wave-level STEPs (2 VALU ops each)
    printf("%-28s %.3f ms  %.2f cycles per wave-STEP per SIMD (2.4 GHz, 1024 SIMDs)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / steps);
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("and + add", d);
    run<1>("and + bcnt(accumulate)", d);
    run<2>("bfe + xor + add", d);
    run<3>("mul_lo + add", d);
    return 0;
}
