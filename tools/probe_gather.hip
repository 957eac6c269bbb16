// probe_gather.hip - what does a gather of 136-byte rows cost as a function of the SPAN the rows are spread over? (diagnostics:
// rescore_kernel's launches take 4-5 x longer for the same number of survivors once a chunk's rows span hundreds of MB.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gather.hip -o tools/bin/probe_gather
// n survivors (rows ascending, evenly spread with jitter over `span` bytes of a 16 GiB buffer), one lane per survivor, the row read
// as rescore_kernel reads it (8 dependent-free 16-byte pieces, one after the other with a little arithmetic between).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
struct __attribute__((aligned(8))) U4 { unsigned x, y, z, w; };
__global__ void __launch_bounds__(256) gather(const char* base, const unsigned long long* off, unsigned n, unsigned* out, int pieces) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const char* p = base + off[i];
        unsigned acc = 0;
        for (int k = 0; k < pieces; k++) {
            const U4 t = *reinterpret_cast<const U4*>(p + 8 + 16 * k);
            acc = acc * 31u + (t.x ^ t.y ^ t.z ^ t.w);
            // ~ the lane-ops of a 128-sample block
#pragma unroll 16
            for (int j = 0; j < 64; j++) acc = acc * 1664525u + 1013904223u;
        }
        if (acc == 0x12345u) out[0] = acc;
    }
}
int main() {
    const size_t bytes = (size_t)16 << 30;
    char* d;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    unsigned* out;
    CK(hipMalloc(&out, 4));
    const unsigned n = 1u << 20;
    unsigned long long* doff;
    CK(hipMalloc(&doff, n * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (double span : {8e6, 64e6, 256e6, 1e9, 4e9, 13.6e9}) {
        std::vector<unsigned long long> off(n);
        const double step = span / n;
        unsigned long long s = 88172645463325252ull;
        for (unsigned i = 0; i < n; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const double jit = (double)(s % 1000) / 1000.0;
            off[i] = (unsigned long long)((i + jit) * step) / 136 * 136;
        }
        // column-major order as the key lists have it: 101 columns' runs, each ascending over the span
        std::vector<unsigned long long> perm(n);
        const unsigned cols = 101, per = n / cols;
        for (unsigned i = 0; i < n; i++) { const unsigned c = i / per < cols ? i / per : cols - 1, j = i - c * per; perm[i] = off[(size_t)((j * (double)cols + c) < n ? (j * cols + c) : i)]; }
        CK(hipMemcpy(doff, perm.data(), n * 8, hipMemcpyHostToDevice));
        for (int pieces : {1, 8}) {
            float best = 1e9f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(gather, dim3(2048), dim3(256), 0, 0, d, doff, n, out, pieces);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("span %8.0f MB, %d piece(s) per row: %7.1f us for %u rows (%.2f G rows/s)\n", span / 1e6, pieces, best * 1e3, n, n / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
