// probe_duplex.hip - how long does a 7 MB device -> pinned-host copy take while 128 MiB host -> device copies run?
// (the streamed ingest: a chunk's record copy was seen to take 2.1 ms beside a piece copy, 0.09 ms alone)
//   hipcc --offload-arch=gfx950 -O2 tools/probe_duplex.hip -o /tmp/probe_duplex && /tmp/probe_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t BIG = 128ull << 20, SMALL = 7ull << 20;
    void *h_big, *d_big, *h_small, *d_small;
    CK(hipHostMalloc(&h_big, BIG, hipHostMallocMapped));
    CK(hipHostMalloc(&h_small, SMALL, hipHostMallocMapped));
    CK(hipMalloc(&d_big, BIG));
    CK(hipMalloc(&d_small, SMALL));
    void *hd_big, *hd_small;
    CK(hipHostGetDevicePointer(&hd_big, h_big, 0));
    CK(hipHostGetDevicePointer(&hd_small, h_small, 0));
    hipStream_t sa, sb;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, greatest));
    // mode of the big H2D: 0 none, 1 one hipMemcpyAsync, 2 sixteen of 8 MiB, 3 copy kernel reading mapped host memory (256 blocks), 4 same with 64 blocks
    // mode of the small D2H: 0 hipMemcpyAsync, 1 copy kernel writing mapped host memory; 2 = 0 + blocking-sync event wait,
    // 3 = 0 + event query polling, 4 = a kernel first (as the scan's chunks), then 0 + blocking event
    hipEvent_t ev_block, ev_plain;
    CK(hipEventCreateWithFlags(&ev_block, hipEventBlockingSync));
    CK(hipEventCreate(&ev_plain));
    for (int big = 0; big <= 2; big++)
        for (int small = 0; small <= 4; small++) {
            double worst = 0, sum = 0, big_ms = 0;
            const int reps = 20;
            for (int r = 0; r < reps + 2; r++) {
                CK(hipDeviceSynchronize());
                const double t0 = now_ms();
                for (int q = 0; q < 2; q++) {  // two pieces queued, as the ingest does
                    if (big == 1) CK(hipMemcpyAsync(d_big, h_big, BIG, hipMemcpyHostToDevice, sa));
                    if (big == 2) for (size_t o = 0; o < BIG; o += 8u << 20) CK(hipMemcpyAsync((char*)d_big + o, (char*)h_big + o, 8u << 20, hipMemcpyHostToDevice, sa));
                    if (big == 3) hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, sa, (const uint4*)hd_big, (uint4*)d_big, BIG / 16);
                    if (big == 4) hipLaunchKernelGGL(copy_kernel, dim3(64), dim3(256), 0, sa, (const uint4*)hd_big, (uint4*)d_big, BIG / 16);
                }
                // a little later: the small copy
                while (now_ms() - t0 < 0.3) {}
                const double t1 = now_ms();
                if (small == 4) hipLaunchKernelGGL(copy_kernel, dim3(128), dim3(256), 0, sb, (const uint4*)d_small, (uint4*)d_big, SMALL / 16);
                if (small != 1) CK(hipMemcpyAsync(h_small, d_small, SMALL, hipMemcpyDeviceToHost, sb));
                else hipLaunchKernelGGL(copy_kernel, dim3(128), dim3(256), 0, sb, (const uint4*)d_small, (uint4*)hd_small, SMALL / 16);
                if (small == 2 || small == 4) { CK(hipEventRecord(ev_block, sb)); CK(hipEventSynchronize(ev_block)); }
                else if (small == 3) { CK(hipEventRecord(ev_plain, sb)); while (hipEventQuery(ev_plain) != hipSuccess) {} }
                else CK(hipStreamSynchronize(sb));
                const double t2 = now_ms();
                CK(hipStreamSynchronize(sa));
                const double t3 = now_ms();
                if (r >= 2) { sum += t2 - t1; if (t2 - t1 > worst) worst = t2 - t1; big_ms += t3 - t0; }
            }
            printf("big H2D mode %d, small D2H mode %d: small copy %.3f ms mean, %.3f worst; two big copies done after %.2f ms (%.1f GB/s)\n", big, small,
                   sum / reps, worst, big_ms / reps, big ? 2.0 * BIG / (big_ms / reps * 1e-3) / 1e9 : 0.0);
        }
    return 0;
}
