// probe_mx32.hip — v_mfma_scale_f32_32x32x64_f8f6f4 against v_mfma_scale_f32_16x16x128_f8f6f4 (FP4 A, FP6 / FP4 B) on a gfx950
// SIMD, alone and with vector-ALU instructions of the same wave between them (diagnostics, not product): would the
// block-scaled filter gain from 32 x 32 tiles (half the instructions, half the B operand reads per multiply-add)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mx32.hip -o tools/bin/probe_mx32
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define M16_6(c, a, b, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define M16_4(c, a, b, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define M32_6(c, a, b, sa, sb) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define M32_4(c, a, b, sa, sb) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define VALU(x, y) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(msk))

// MODE 0: 16x16x128, 28 accumulators (112 registers); MODE 1: 32x32x64, 7 accumulators (112 registers).
// V2 / 2 = vector-ALU instructions per 32 768 multiply-adds (= per 16x16x128 instruction, per HALF a 32x32x64 instruction).
template <int MODE, int V2>
__global__ void __launch_bounds__(512) k(float* out, int iters, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63u;
    v4i A[4], B4[2];
    v6i B6[2];
#pragma unroll
    for (int i = 0; i < 4; i++) A[i] = (v4i){(int)(0x11111111u & (lane * 0x01010101u + seed)), 0x22222222 & (int)seed, 0x11110000, 0x00001111};
    B6[0] = B6[1] = (v6i){0x08208208, 0x20820820, (int)0x82082082u, 0x08208208, 0x20820820, (int)0x82082082u};
    B4[0] = B4[1] = (v4i){0x12341234, 0x21212121, 0x11111111, 0x22222222};
    uint32_t msk = seed | 0x11111111u, x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = lane + i, y[i] = lane * (i + 3);
    const int sa = 0x7F7F7F7F, sb = 0x84848484;
    float s = 0;
    if (MODE == 0) {
        v4f acc[28];
#pragma unroll
        for (int i = 0; i < 28; i++) acc[i] = (v4f){0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 56; i++) {
                if (i & 1) M16_4(acc[i >> 1], A[i & 3], B4[(i >> 2) & 1], sa, sa);
                else M16_6(acc[i >> 1], A[i & 3], B6[(i >> 2) & 1], sa, sb);
#pragma unroll
                for (int v = 0; v < (V2 + 1) / 2; v++)
                    if (2 * v < V2 || (i & 1)) VALU(x[(i + v) & 7], y[(i * 3 + v) & 7]);
            }
        }
#pragma unroll
        for (int i = 0; i < 28; i++) s += acc[i][0] + acc[i][3];
    } else {
        v16f acc[7];
#pragma unroll
        for (int i = 0; i < 7; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[i][j] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 28; i++) {  // 28 instructions = the multiply-adds of 56 of the small ones
                if (i & 1) M32_4(acc[(i >> 1) % 7], A[i & 3], B4[(i >> 2) & 1], sa, sa);
                else M32_6(acc[(i >> 1) % 7], A[i & 3], B6[(i >> 2) & 1], sa, sb);
#pragma unroll
                for (int v = 0; v < V2; v++) VALU(x[(i + v) & 7], y[(i * 3 + v) & 7]);
            }
        }
#pragma unroll
        for (int i = 0; i < 7; i++) s += acc[i][0] + acc[i][15];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)(x[i] & 1u);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int V2>
static void run(const char* name, float* d, int threads) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, V2>), dim3(256), dim3(threads), 0, 0, d, iters, 0x01010101u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double units = (double)iters * 56.0 * (threads / 64) * 256.0;  // 65 536-multiply-add units on the chip
    printf("%-58s %d wave(s)/SIMD: %8.3f ms  %7.1f TOP/s  %5.2f ns per unit and SIMD\n", name, threads / 256, best, units * 65536.0 / (best * 1e-3) * 1e-12,
           best * 1e6 / ((double)iters * 56.0 * (threads / 256)));
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<0, 0>("16x16x128 fp4 x fp6/fp4 alone", d, threads);
        run<1, 0>("32x32x64  fp4 x fp6/fp4 alone", d, threads);
        run<0, 2>("16x16x128 + 1 VALU per unit", d, threads);
        run<1, 2>("32x32x64  + 1 VALU per unit (2 per instruction)", d, threads);
        run<0, 4>("16x16x128 + 2 VALU per unit", d, threads);
        run<1, 4>("32x32x64  + 2 VALU per unit (4 per instruction)", d, threads);
        run<0, 6>("16x16x128 + 3 VALU per unit", d, threads);
        run<1, 6>("32x32x64  + 3 VALU per unit (6 per instruction)", d, threads);
    }
    return 0;
}
