// probe_mx.hip — operand layout and issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 with an FP4 (E2M1) A operand made
// from table bits and an FP8 (E4M3) / FP6 (E2M3) B operand holding small integers (diagnostics, not product).
//   build: hipcc --offload-arch=gfx950 -O3 tools/probe_mx.hip -o tools/bin/probe_mx
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const uint32_t* A, const uint32_t* B, float* C) {
    const int lane = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) a[i] = A[lane * 4 + i];
    for (int i = 0; i < 8; i++) b[i] = B[lane * 8 + i];
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 4; i++) C[lane * 4 + i] = c[i];
}

template <int MODE>  // 0: fp4 x fp8, 1: fp4 x fp6, 2: int8 16x16x64, 3: fp4 x fp4
__global__ void rate_kernel(float* out, int iters) {
    v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {5, 6, 7, (int)threadIdx.x, 1, 2, 3, 4};
    v4f c[8];
    v4i ci[8];
    for (int i = 0; i < 8; i++) {
        c[i] = (v4f){0, 0, 0, 0};
        ci[i] = (v4i){0, 0, 0, 0};
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 4, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 1) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 4, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 3) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 2) ci[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i){a[0], a[1], a[2], a[3]}, (v4i){b[0], b[1], b[2], b[3]}, ci[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + (float)ci[i][0] + (float)ci[i][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint8_t e4m3_of_int(int v) {  // |v| <= 15
    if (v == 0) return 0;
    const int s = v < 0 ? 0x80 : 0, a = std::abs(v);
    int e = 0;
    while ((2 << e) <= a) e++;
    const int mant = (a * 8) / (1 << e) - 8;
    return (uint8_t)(s | ((e + 7) << 3) | mant);
}

int main() {
    std::mt19937 rng(5);
    std::vector<uint8_t> Abit(16 * 128);
    std::vector<int> Bv(128 * 16);
    for (auto& x : Abit) x = rng() & 1;
    for (auto& x : Bv) x = (int)(rng() % 31) - 15;
    for (int nib = 1; nib <= 2; nib++) {
        std::vector<uint32_t> A(64 * 4, 0), B(64 * 8, 0);
        for (int lane = 0; lane < 64; lane++) {
            const int m = lane & 15, kb = lane >> 4;
            for (int e = 0; e < 32; e++) {
                if (Abit[m * 128 + 32 * kb + e]) A[lane * 4 + e / 8] |= (uint32_t)nib << (4 * (e % 8));
                B[lane * 8 + e / 4] |= (uint32_t)e4m3_of_int(Bv[(32 * kb + e) * 16 + m]) << (8 * (e % 4));
            }
        }
        uint32_t *dA, *dB;
        float* dC;
        hipMalloc(&dA, A.size() * 4);
        hipMalloc(&dB, B.size() * 4);
        hipMalloc(&dC, 64 * 4 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int i = 0; i < 4; i++) {
                const int col = lane & 15, row = (lane >> 4) * 4 + i;
                double exp = 0;
                for (int k = 0; k < 128; k++) exp += Abit[row * 128 + k] * (nib == 1 ? 0.5 : 1.0) * Bv[k * 16 + col];
                if (std::fabs(exp - C[lane * 4 + i]) > 1e-6) {
                    if (bad < 5) printf("nib %d mismatch lane %d i %d: got %g exp %g\n", nib, lane, i, C[lane * 4 + i], exp);
                    bad++;
                }
            }
        printf("layout check (A nibble 0x%d): %s (%d mismatches of 256)\n", nib, bad ? "FAIL" : "ok", bad);
    }
    float* dout;
    hipMalloc(&dout, 1024 * 256 * 4);
    const int iters = 20000;
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(256), 0, 0, dout, iters);
            if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(256), 0, 0, dout, iters);
            if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(1024), dim3(256), 0, 0, dout, iters);
            if (mode == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(1024), dim3(256), 0, 0, dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = 1024.0 * 4 * iters * 8;  // per wave: iters * 8
        const double macs = n_mfma * 16 * 16 * (mode == 2 ? 64 : 128);
        printf("mode %d (%s): %.3f ms, %.1f T MAC/s = %.0f TOP/s\n", mode,
               mode == 0 ? "fp4 x fp8 16x16x128" : mode == 1 ? "fp4 x fp6 16x16x128" : mode == 2 ? "i8 16x16x64" : "fp4 x fp4 16x16x128", ms,
               macs / (ms * 1e-3) / 1e12, 2 * macs / (ms * 1e-3) / 1e12);
    }
    return 0;
}
