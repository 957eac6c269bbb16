// probe_mx_power.hip — what the matrix pipe sustains for SECONDS, with the board's power and shader clock read beside it
// (diagnostics, not product; verdict r05 item 1b: is mx_kernel's ~2.0 GHz a power limit or an issue limit?).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mx_power.hip -o tools/bin/probe_mx_power -pthread
// Streams of v_mfma_scale_f32_16x16x128_f8f6f4 exactly as the filter issues them (8 FP4 x FP6 + 8 FP4 x FP4 per unit of 16, 28
// accumulators), two waves per SIMD on every CU, each variant run back to back for ~3 s:
//   zeros     all operands zero (the pipe's floor: issue and clock without data toggling)
//   sparseA   A = table-like nibbles (one of bits 0..2 set at density 1/2), B = random FP6 / FP4 codes: the filter's data
//   denseA    A and B random bit patterns
//   sparseA+V the same with 8 / 16 / 24 independent v_and_b32 per 16 MFMAs (the filter's main loop has ~8, with its epilogue ~24)
//   ... + L   and six ds_read_b128 per 16 MFMAs (the operand pieces the filter reads from LDS), waited for one unit later
// For each: MFMAs per second and SIMD, the implied pipe-busy fraction at the MEASURED mean clock, mean / max socket power, cap.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
#include <limits.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define MFMA6(c, a, b, sa, sb) \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define MFMA4(c, a, b, sa, sb) \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define VALU(x, y) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(msk))
#define LDSR(b, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(off))

__device__ inline uint32_t mixu(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// DATA: 0 zeros, 1 sparse A + random B, 2 dense A + random B.   V: independent VALU per unit of 16 MFMAs
// L: ds_read_b128 per unit (the filter reads six operand pieces per unit; waited for one unit later, as there)
template <int DATA, int V, int L = 0>
__global__ void __launch_bounds__(512) k(float* out, int passes, uint32_t seed) {
    __shared__ v4i lds[L ? 4096 : 1];
    const uint32_t lane = threadIdx.x & 63u, gid = blockIdx.x * 512u + threadIdx.x;
    if (L) {
        for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) lds[i] = (v4i){(int)mixu(i + seed), (int)mixu(i * 3u), (int)mixu(i * 5u), (int)mixu(i * 7u)};
        __syncthreads();
    }
    const uint32_t laddr = (uint32_t)(size_t)lds + lane * 16u;
    v4i ld[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ld[i] = (v4i){0, 0, 0, 0};
    v4f acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = (v4f){0, 0, 0, 0};
    v4i A[4], B4[2];
    v6i B6[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = mixu(gid * 16u + i * 4u + q + seed);
            w[q] = DATA == 0 ? 0u : DATA == 1 ? (r & 0x11111111u) : r;
        }
        A[i] = (v4i){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        uint32_t w[10];
#pragma unroll
        for (int q = 0; q < 10; q++) w[q] = DATA == 0 ? 0u : mixu(gid * 32u + i * 10u + q + 77u + seed);
        B6[i] = (v6i){(int)w[0], (int)w[1], (int)w[2], (int)w[3], (int)w[4], (int)w[5]};
        B4[i] = (v4i){(int)w[6], (int)w[7], (int)w[8], (int)w[9]};
    }
    const int s0 = 0x7F7F7F7F;  // block scale 2^0 (the accumulators may overflow to inf / NaN: irrelevant here)
    uint32_t msk = 0x11111111u, x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = lane + i;
    for (int p = 0; p < passes; p++) {
#pragma unroll
        for (int u = 0; u < 28; u++) {  // a unit: two column tiles x four row tiles x two slices
            if (L > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int l = 0; l < L; l++) LDSR(ld[l & 7], laddr, ((u * 8 + l) & 63) * 1024);
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) MFMA6(acc[(u * 8 + t * 4 + r) % 28], A[r], B6[t], s0, s0);
#pragma unroll
                for (int r = 0; r < 4; r++) MFMA4(acc[(u * 8 + t * 4 + r) % 28], A[r], B4[t], s0, s0);
            }
#pragma unroll
            for (int v = 0; v < V; v++) VALU(x[v % 8], x[(v + 3) % 8]);
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 28; i++) r += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; i++) r += (float)x[i] + (float)(ld[i][0] & 1);
    if (r == 12345.678f) out[gid] = r;
}

struct Sensors {
    std::string power, freq, cap;
};
static long read_long(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    long v = -1;
    if (fscanf(f, "%ld", &v) != 1) v = -1;
    fclose(f);
    return v;
}
static Sensors find_sensors(const char* pci) {  // the hwmon directory of the card whose PCI address is `pci` (lower case)
    Sensors s;
    DIR* d = opendir("/sys/class/drm");
    if (!d) return s;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
        const std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
        char real[PATH_MAX];
        if (!realpath(dev.c_str(), real)) continue;
        std::string r(real);
        for (auto& c : r) c = (char)tolower(c);
        if (r.find(pci) == std::string::npos) continue;
        DIR* h = opendir((dev + "/hwmon").c_str());
        if (!h) continue;
        while (dirent* he = readdir(h)) {
            if (strncmp(he->d_name, "hwmon", 5) != 0) continue;
            const std::string b = dev + "/hwmon/" + he->d_name + "/";
            s.power = read_long(b + "power1_average") >= 0 ? b + "power1_average" : b + "power1_input";
            s.freq = b + "freq1_input";
            s.cap = b + "power1_cap";
        }
        closedir(h);
    }
    closedir(d);
    return s;
}

template <int DATA, int V, int L = 0>
static void run(const char* name, float* d, const Sensors& sn, double seconds) {
    const int passes = 4000;  // 448 MFMAs per pass and wave: ~30 ms per launch
    std::atomic<bool> stop{false};
    double p_sum = 0, p_max = 0, f_sum = 0, f_min = 1e30, f_max = 0;
    long n = 0;
    hipLaunchKernelGGL((k<DATA, V, L>), dim3(256), dim3(512), 0, 0, d, 100, 1u);
    hipDeviceSynchronize();
    std::thread th([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(500));  // let the clock settle under the load
        while (!stop.load()) {
            const long p = read_long(sn.power), f = read_long(sn.freq);
            if (p >= 0 && f >= 0) {
                p_sum += p * 1e-6; p_max = std::max(p_max, p * 1e-6);
                f_sum += f * 1e-6; f_min = std::min(f_min, f * 1e-6); f_max = std::max(f_max, f * 1e-6);
                n++;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    double gpu_ms = 0;
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int i = 0; i < 4; i++) hipLaunchKernelGGL((k<DATA, V, L>), dim3(256), dim3(512), 0, 0, d, passes, (uint32_t)launches);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        gpu_ms += ms;
        launches += 4;
    }
    stop.store(true);
    th.join();
    const double mfma_per_simd = (double)launches * passes * 448.0 * 2.0;  // two waves per SIMD
    const double ns_per = gpu_ms * 1e6 / mfma_per_simd;
    const double f = n ? f_sum / n : 0;
    printf("%-44s %6.2f ns/MFMA/SIMD  %5.2f POP/s  clock %4.0f MHz (%4.0f-%4.0f)  busy@16cyc %.3f  power %4.0f W (max %4.0f)  cap %ld W\n", name, ns_per,
           256.0 * 4 * 65536.0 / ns_per * 1e-6, f, f_min, f_max, f > 0 ? 16.0 / (ns_per * f * 1e-3) : 0.0, n ? p_sum / n : 0.0, p_max,
           read_long(sn.cap) / 1000000);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    char pci[64] = {0};
    hipDeviceGetPCIBusId(pci, sizeof(pci), 0);
    for (char* c = pci; *c; c++) *c = (char)tolower(*c);
    const Sensors sn = find_sensors(pci);
    printf("device %s  power sensor %s\n", pci, sn.power.c_str());
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    run<0, 0>("zeros", d, sn, secs);
    run<1, 0>("sparse A, random B (the filter's data)", d, sn, secs);
    run<2, 0>("dense A, random B", d, sn, secs);
    run<1, 8>("sparse A + 8 VALU per 16 MFMAs", d, sn, secs);
    run<1, 16>("sparse A + 16 VALU per 16 MFMAs", d, sn, secs);
    run<1, 24>("sparse A + 24 VALU per 16 MFMAs", d, sn, secs);
    run<1, 8, 6>("sparse A + 8 VALU + 6 ds_read_b128 per 16 MFMAs", d, sn, secs);
    run<1, 24, 6>("sparse A + 24 VALU + 6 ds_read_b128 per 16", d, sn, secs);
    return 0;
}
