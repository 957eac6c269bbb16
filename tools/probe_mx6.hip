// probe_mx6.hip — v_mfma_scale_f32_16x16x128_f8f6f4 with an FP4 (E2M1) table-bit A operand and FP6 (E2M3) / FP4 slice
// B operands accumulated into ONE accumulator through the block scales (diagnostics, not product):
//   1. where element e of a lane's FP6 operand sits in K, and whether E2M3 subnormals (1/8 .. 7/8) multiply exactly;
//   2. that scale_b = 2^s on slice 0 and 2^0 on slice 1 gives acc = 0.5 * sum g (2^s v0 + v1) exactly;
//   3. issue rate and sustained clock of the interleaved FP4xFP6 / FP4xFP4 stream.
//   build: hipcc --offload-arch=gfx950 -O3 tools/probe_mx6.hip -o tools/bin/probe_mx6
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// A: [64 lanes][4 dwords] FP4, B0: [64][6] FP6, B1: [64][4] FP4 or [64][6] FP6 (fmt1 = 4 / 2)
template <int FMT1>
__global__ void layout_kernel(const uint32_t* A, const uint32_t* B0, const uint32_t* B1, float* C, int sa, int sb0, int sb1) {
    const int lane = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b0 = {0, 0, 0, 0, 0, 0, 0, 0}, b1 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) a[i] = A[lane * 4 + i];
    for (int i = 0; i < 6; i++) b0[i] = B0[lane * 6 + i];
    for (int i = 0; i < (FMT1 == 4 ? 4 : 6); i++) b1[i] = B1[lane * 6 + i];
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b0, c, 4, 2, 0, sa, 0, sb0);
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b1, c, 4, FMT1, 0, sa, 0, sb1);
    for (int i = 0; i < 4; i++) C[lane * 4 + i] = c[i];
}

template <int MODE>  // 0: fp4 x fp6 only, 1: fp4 x fp6 / fp4 x fp4 interleaved, 2: fp4 x fp4 only, 3: i8 16x16x64
__global__ void rate_kernel(float* out, long long* clk, int iters) {
    v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {5, 6, 7, (int)threadIdx.x, 1, 2, 0, 0}, b4 = {5, 6, 7, (int)threadIdx.x, 0, 0, 0, 0};
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4f c[8];
    v4i ci[8];
    for (int i = 0; i < 8; i++) {
        c[i] = (v4f){0, 0, 0, 0};
        ci[i] = (v4i){0, 0, 0, 0};
    }
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 4, 2, 0, 0x7F7F7F7F, 0, 0x82828282);
            if (MODE == 1 && (i & 1) == 0) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 4, 2, 0, 0x7F7F7F7F, 0, 0x82828282);
            if (MODE == 1 && (i & 1) == 1) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b4, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 2) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b4, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 3) ci[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8((v4i){a[0], a[1], a[2], a[3]}, (v4i){b[0], b[1], b[2], b[3]}, ci[i], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + (float)ci[i][0] + (float)ci[i][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

// E2M3: sign(1) exp(2) mant(3), bias 1; q = value * 8 must be in {0..15, 16..30 step 2, 32..60 step 4}
static uint32_t e2m3_of_q(int q) {
    const uint32_t s = q < 0 ? 0x20u : 0u;
    const int a = std::abs(q);
    if (a < 8) return s | (uint32_t)a;  // subnormal: m / 8
    int e = 1, base = 8;
    while (a >= 2 * base) {
        base *= 2;
        e++;
    }
    const int step = base / 8;
    if (a % step) {
        printf("bad q %d\n", q);
        exit(1);
    }
    return s | ((uint32_t)e << 3) | (uint32_t)((a - base) / step);
}
// E2M1 in units of 0.5: {0,1,2,3,4,6,8,12}
static uint32_t e2m1_of_h(int h) {
    const uint32_t s = h < 0 ? 8u : 0u;
    static const int tab[8] = {0, 1, 2, 3, 4, 6, 8, 12};
    for (uint32_t i = 0; i < 8; i++)
        if (tab[i] == std::abs(h)) return s | i;
    printf("bad h %d\n", h);
    exit(1);
}

int main() {
    std::mt19937 rng(11);
    std::vector<int> g6, g4 = {0, 1, 2, 3, 4, 6, 8, 12};
    for (int q = 0; q < 16; q++) g6.push_back(q);
    for (int q = 16; q <= 30; q += 2) g6.push_back(q);
    for (int q = 32; q <= 60; q += 4) g6.push_back(q);
    uint32_t *dA, *dB0, *dB1;
    float* dC;
    hipMalloc(&dA, 64 * 4 * 4);
    hipMalloc(&dB0, 64 * 6 * 4);
    hipMalloc(&dB1, 64 * 6 * 4);
    hipMalloc(&dC, 64 * 4 * 4);
    for (int fmt1 : {2, 4}) {
        for (int hyp = 0; hyp < 2; hyp++) {      // 0: k = 32 kb + e   1: k = 16 kb + e (e < 16), 64 + 16 kb + e - 16
            for (int nib = 0; nib < 3; nib++) {  // which nibble bit carries the table bit (scale_a 2^-nib)
                std::vector<uint8_t> Abit(16 * 128);
                std::vector<int> Q0(128 * 16), Q1(128 * 16);
                for (auto& x : Abit) x = rng() & 1;
                for (auto& x : Q0) x = g6[rng() % g6.size()] * ((rng() & 1) ? -1 : 1);
                for (auto& x : Q1) x = (fmt1 == 2 ? g6[rng() % g6.size()] : g4[rng() % g4.size()]) * ((rng() & 1) ? -1 : 1);
                std::vector<uint32_t> A(64 * 4, 0), B0(64 * 6, 0), B1(64 * 6, 0);
                for (int lane = 0; lane < 64; lane++) {
                    const int m = lane & 15, kb = lane >> 4;
                    for (int e = 0; e < 32; e++) {
                        if (Abit[m * 128 + 32 * kb + e]) A[lane * 4 + e / 8] |= (uint32_t)(1u << nib) << (4 * (e % 8));
                        const int k = hyp == 0 ? 32 * kb + e : (e < 16 ? 16 * kb + e : 64 + 16 * kb + e - 16);
                        const uint64_t f0 = e2m3_of_q(Q0[k * 16 + m]);
                        const int bit = 6 * e;
                        B0[lane * 6 + bit / 32] |= (uint32_t)(f0 << (bit % 32));
                        if (bit % 32 > 26) B0[lane * 6 + bit / 32 + 1] |= (uint32_t)(f0 >> (32 - bit % 32));
                        // the FP4 operand's map is known (probe_mx): k = 32 kb + e
                        if (fmt1 == 2) {
                            const uint64_t f1 = e2m3_of_q(Q1[k * 16 + m]);
                            B1[lane * 6 + bit / 32] |= (uint32_t)(f1 << (bit % 32));
                            if (bit % 32 > 26) B1[lane * 6 + bit / 32 + 1] |= (uint32_t)(f1 >> (32 - bit % 32));
                        } else {
                            B1[lane * 6 + e / 8] |= e2m1_of_h(Q1[(32 * kb + e) * 16 + m]) << (4 * (e % 8));
                        }
                    }
                }
                hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
                hipMemcpy(dB0, B0.data(), B0.size() * 4, hipMemcpyHostToDevice);
                hipMemcpy(dB1, B1.data(), B1.size() * 4, hipMemcpyHostToDevice);
                const int s = fmt1 == 2 ? 5 : 3;
                const int sa = 0x01010101 * (0x7F - nib), sb0 = 0x01010101 * (0x7F + s), sb1 = 0x7F7F7F7F;
                if (fmt1 == 2)
                    hipLaunchKernelGGL(layout_kernel<2>, dim3(1), dim3(64), 0, 0, dA, dB0, dB1, dC, sa, sb0, sb1);
                else
                    hipLaunchKernelGGL(layout_kernel<4>, dim3(1), dim3(64), 0, 0, dA, dB0, dB1, dC, sa, sb0, sb1);
                std::vector<float> C(256);
                hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
                int bad = 0;
                for (int lane = 0; lane < 64; lane++)
                    for (int i = 0; i < 4; i++) {
                        const int col = lane & 15, row = (lane >> 4) * 4 + i;
                        double exp = 0;
                        for (int k = 0; k < 128; k++)
                            exp += Abit[row * 128 + k] * 0.5 *
                                   ((double)(1 << s) * Q0[k * 16 + col] / 8.0 + (fmt1 == 2 ? Q1[k * 16 + col] / 8.0 : Q1[k * 16 + col] * 0.5));
                        if (exp != (double)C[lane * 4 + i]) {
                            if (bad < 3) printf("  mismatch lane %d i %d: got %.6f exp %.6f\n", lane, i, C[lane * 4 + i], exp);
                            bad++;
                        }
                    }
                printf("slice1 %s, fp6 map hypothesis %d, table bit in nibble bit %d: %s (%d mismatches of 256)\n", fmt1 == 2 ? "fp6" : "fp4",
                       hyp, nib, bad ? "FAIL" : "ok", bad);
            }
        }
    }
    float* dout;
    long long* dclk;
    hipMalloc(&dout, 1024 * 256 * 4);
    hipMalloc(&dclk, 16);
    const int iters = 20000;
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(256), 0, 0, dout, dclk, iters);
            if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(256), 0, 0, dout, dclk, iters);
            if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(1024), dim3(256), 0, 0, dout, dclk, iters);
            if (mode == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(1024), dim3(256), 0, 0, dout, dclk, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long clk[2];
        hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
        const double n_mfma = 1024.0 * 4 * iters * 8;
        const double macs = n_mfma * 16 * 16 * (mode == 3 ? 64 : 128);
        printf("mode %d (%s): %.3f ms, %.0f TOP/s; block 0: %.1f clock64 ticks per MFMA, wall_clock64 %.3f ms\n", mode,
               mode == 0 ? "fp4 x fp6" : mode == 1 ? "fp4 x fp6 / fp4 x fp4 interleaved" : mode == 2 ? "fp4 x fp4" : "i8 16x16x64", ms,
               2 * macs / (ms * 1e-3) / 1e12, (double)clk[0] / (iters * 8.0), (double)clk[1] / 1e5);
    }
    return 0;
}
