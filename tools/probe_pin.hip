// probe_pin.hip - what does pinning host memory cost, and does a huge-page backing make it cheaper?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_pin.hip -o /tmp/probe_pin -pthread && /tmp/probe_pin
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char* p, size_t n, int threads) {
    std::vector<std::thread> t;
    for (int i = 0; i < threads; i++)
        t.emplace_back([=] {
            const size_t a = n / threads * i, b = i == threads - 1 ? n : n / threads * (i + 1);
            for (size_t o = a; o < b; o += 4096) p[o] = 0;
        });
    for (auto& x : t) x.join();
}
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[256] = "?";
    if (f) { fgets(line, sizeof line, f); fclose(f); }
    printf("THP: %s", line);
    const size_t N = 512ull << 20;
    for (int rep = 0; rep < 2; rep++) {
        void* p = nullptr;
        double t0 = now_ms();
        CK(hipHostMalloc(&p, N, hipHostMallocMapped));
        printf("hipHostMalloc(512 MiB, mapped): %.1f ms\n", now_ms() - t0);
        t0 = now_ms();
        CK(hipHostFree(p));
        printf("  hipHostFree: %.1f ms\n", now_ms() - t0);
        for (int huge = 0; huge <= 1; huge++)
            for (int threads : {1, 8}) {
                t0 = now_ms();
                char* m = (char*)mmap(nullptr, N + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                char* al = (char*)(((uintptr_t)m + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
                if (huge) madvise(al, N, MADV_HUGEPAGE);
                touch(al, N, threads);
                const double t1 = now_ms();
                CK(hipHostRegister(al, N, hipHostRegisterMapped));
                const double t2 = now_ms();
                void* d = nullptr;
                CK(hipHostGetDevicePointer(&d, al, 0));
                printf("mmap%s + touch(%d threads) %.1f ms + hipHostRegister %.1f ms = %.1f ms\n", huge ? " + MADV_HUGEPAGE" : "", threads, t1 - t0, t2 - t1, t2 - t0);
                t0 = now_ms();
                CK(hipHostUnregister(al));
                munmap(m, N + (2 << 20));
                printf("  unregister + munmap: %.1f ms\n", now_ms() - t0);
            }
    }
    return 0;
}
