// probe_hbm_read.hip - what does the memory system deliver to a pure read kernel? (the ceiling the one-column filter is
// priced against in DESIGN 4.6).   hipcc --offload-arch=gfx950 -O3 tools/probe_hbm_read.hip -o /tmp/p && /tmp/p [GiB=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ uint4 ld(const uint4* p, bool nt) {
    if (!nt) return *p;
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT, int UNROLL>
__global__ void __launch_bounds__(256) read_stride(const uint4* __restrict__ p, size_t n16, unsigned long long* out) {
    // grid-stride, UNROLL independent 16-byte loads per thread and turn
    const size_t step = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * step < n16; i += UNROLL * step) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = ld(p + i + u * step, NT);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}

template <int NT>
__global__ void __launch_bounds__(256) read_chunks(const uint4* __restrict__ p, size_t n16, unsigned long long* out) {
    // each wave owns contiguous 8 KB pieces (the narrow filter's access pattern: 64 rows x 136 B), lane-linear 1 KB loads
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (size_t c = wave; (c + 1) * 512 <= n16; c += n_waves) {
        const uint4* q = p + c * 512 + lane;
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ld(q + u * 64, NT);
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}

int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 64;
    const size_t bytes = gib << 30, n16 = bytes / 16;
    void* d;
    unsigned long long* out;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(d, 1, bytes));
    CK(hipMemset(out, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("%-44s %8.3f ms  %6.3f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    };
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
        char nm[96];
        snprintf(nm, sizeof nm, "stride x4, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL((read_stride<0, 4>), dim3(blocks), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "stride x4 nontemporal, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL((read_stride<1, 4>), dim3(blocks), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "stride x8, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL((read_stride<0, 8>), dim3(blocks), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "8 KB per wave, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL((read_chunks<0>), dim3(blocks), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "8 KB per wave nontemporal, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL((read_chunks<1>), dim3(blocks), dim3(256), 0, 0, (const uint4*)d, n16, out); });
    }
    return 0;
}
