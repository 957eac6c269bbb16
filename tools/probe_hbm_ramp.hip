// probe_hbm_ramp.hip - does a pure read stream reach its rate at once after the GPU has been idle? (diagnostics; DESIGN 4.6:
// why a one-column pass over 13.6 GB runs at 0.48 of the roofline while the same kernel reaches 0.77 on 163 GB.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_hbm_ramp.hip -o tools/bin/probe_hbm_ramp
// After `idle_ms` of idleness: back-to-back read launches over consecutive pieces of a 16 GiB buffer, sizes growing x3.46 from
// 4 MB (the chunk plan of a one-column scan), each timed with events; then the same pieces again at once (warm clocks).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// 8 KB per wave and turn, lane-linear, non-temporal: how the narrow filter reads its rows
__global__ void __launch_bounds__(256) read_chunks(const uint4* __restrict__ p, size_t n16, unsigned long long* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    unsigned acc = 0;
    for (size_t base = wave * 512; base + 512 <= n16; base += n_waves * 512) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + base + u * 64 + lane));
            v[u] = make_uint4(t.x, t.y, t.z, t.w);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}

int main(int argc, char** argv) {
    const int idle_ms = argc > 1 ? atoi(argv[1]) : 20;
    const size_t bytes = (size_t)16 << 30;
    void* d;
    unsigned long long* out;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(d, 1, bytes));
    CK(hipMemset(out, 0, 8));
    std::vector<size_t> sizes;
    size_t off = 0;
    for (double s = 4e6; off + (size_t)s < (size_t)13.6e9; s *= 3.46) {
        sizes.push_back(((size_t)s) / 8192 * 8192);
        off += sizes.back();
    }
    sizes.push_back(((size_t)13.6e9 - off) / 8192 * 8192);
    std::vector<hipEvent_t> ev(sizes.size() + 1);
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (int round = 0; round < 3; round++) {
        const bool after_idle = round != 1;
        CK(hipDeviceSynchronize());
        if (after_idle) std::this_thread::sleep_for(std::chrono::milliseconds(idle_ms));
        size_t o = 0;
        CK(hipEventRecord(ev[0]));
        for (size_t i = 0; i < sizes.size(); i++) {
            const size_t n16 = sizes[i] / 16;
            const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>(n16 / 512 / 4, 1), 8192);
            hipLaunchKernelGGL(read_chunks, dim3(blocks), dim3(256), 0, 0, (const uint4*)((const char*)d + o), n16, out);
            CK(hipEventRecord(ev[i + 1]));
            o += sizes[i];
        }
        CK(hipDeviceSynchronize());
        printf("%s:\n", after_idle ? (round == 0 ? "after idle" : "after idle (again)") : "at once after the previous round");
        float total = 0;
        for (size_t i = 0; i < sizes.size(); i++) {
            float ms;
            CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            total += ms;
            printf("  piece %2zu %9.1f MB  %8.1f us  %6.2f TB/s\n", i, sizes[i] / 1e6, ms * 1e3, sizes[i] / (ms * 1e-3) / 1e12);
        }
        printf("  all %.1f MB in %.3f ms = %.2f TB/s\n", o / 1e6, total, o / (total * 1e-3) / 1e12);
    }
    return 0;
}
