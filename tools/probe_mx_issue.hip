// probe_mx_issue.hip — how v_mfma_scale_f32_16x16x128_f8f6f4 shares a gfx950 SIMD with vector-ALU instructions and LDS
// reads issued by the SAME wave (one wave per SIMD) or by two waves per SIMD (diagnostics, not product). It answers one
// question for score_mx.hip: can ONE wave per SIMD with two accumulator sets run the epilogue of pass i between the MFMAs
// of pass i + 1 (interleaved in its own instruction stream), where two waves per SIMD do not hide each other's epilogues?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mx_issue.hip -o tools/bin/probe_mx_issue
// A "unit" is 16 MFMAs (8 FP4 x FP6 + 8 FP4 x FP4, 16 independent accumulators of 28); a pass is 28 units (448 MFMAs).
//   interleaved<V, L>: every unit carries V independent VALU instructions and L ds_read_b128 (waited for one unit later)
//   phased<V, L, E>  : the same units, then E VALU instructions alone per pass (the epilogue as it runs today)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define MFMA6(c, a, b, sa, sb) \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define MFMA4(c, a, b, sa, sb) \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define VALU(x, y) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(msk))
#define LDSR(b, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(off))

template <int V, int L, int E>
__global__ void __launch_bounds__(512) k(float* out, int passes, uint32_t seed) {
    __shared__ v4i lds[64 * 64];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < 64u * 64u; i += blockDim.x) lds[i] = (v4i){(int)(i & 3), 1, 2, 1};
    __syncthreads();
    v4f acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = (v4f){0, 0, 0, 0};
    v4i A[4], B4[2];
    v6i B6[2];
#pragma unroll
    for (int i = 0; i < 4; i++) A[i] = (v4i){(int)(0x11111111u & (lane * 0x01010101u + seed)), 0x22222222 & (int)seed, 0x11110000, 0x00001111};
    B6[0] = B6[1] = (v6i){0x08208208, 0x20820820, (int)0x82082082u, 0x08208208, 0x20820820, (int)0x82082082u};
    B4[0] = B4[1] = (v4i){0x12341234, 0x21212121, 0x11111111, 0x22222222};
    v4i ld[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ld[i] = (v4i){0, 0, 0, 0};
    uint32_t msk = seed | 0x11111111u, x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = lane + i, y[i] = lane * (i + 3);
    const int sa = 0x7F7F7F7F, sb = 0x84848484;
    const uint32_t laddr = (uint32_t)(size_t)lds + lane * 16u;
    for (int p = 0; p < passes; p++) {
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int u = 0; u < 14; u++) {
                // LDS reads of this unit (consumed - waited for - at the start of the next one)
                if (L > 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int l = 0; l < L; l++) LDSR(ld[l & 7], laddr, ((u * 8 + l) & 63) * 1024);
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int ai = (u * 2 + (i >> 3)) % 28;
                    if (i < 8) MFMA6(acc[(u * 16 + i) % 28], A[i & 3], B6[i & 1], sa, sb);
                    else MFMA4(acc[(u * 16 + i) % 28], A[i & 3], B4[i & 1], sa, sa);
                    (void)ai;
                    // V VALU instructions spread over the unit's 16 MFMAs
                    constexpr int per = (V + 15) / 16;
#pragma unroll
                    for (int v = 0; v < per; v++)
                        if (i * per + v < V) VALU(x[(i + v) & 7], y[(i * 3 + v) & 7]);
                }
            }
        }
        if (E > 0) {
#pragma unroll 8
            for (int e = 0; e < E; e++) VALU(x[e & 7], y[(e + 3) & 7]);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 28; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; i++) s += (float)(x[i] & 1u) + (float)(ld[i][0] & 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int L, int E>
static void run(const char* name, float* d, int threads) {
    const int passes = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, L, E>), dim3(256), dim3(threads), 0, 0, d, passes, 0x01010101u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfma_per_simd = (double)passes * 448.0 * (threads / 256);
    printf("%-64s %d wave(s)/SIMD: %8.3f ms  %6.2f ns per MFMA per SIMD  (%.1f cycles at 2.0 GHz)\n", name, threads / 256, best,
           best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.0);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<0, 0, 0>("MFMA only", d, threads);
        run<8, 0, 0>("interleaved: 8 VALU per 16 MFMAs", d, threads);
        run<16, 0, 0>("interleaved: 16 VALU per 16 MFMAs", d, threads);
        run<24, 0, 0>("interleaved: 24 VALU per 16 MFMAs", d, threads);
        run<32, 0, 0>("interleaved: 32 VALU per 16 MFMAs", d, threads);
        run<48, 0, 0>("interleaved: 48 VALU per 16 MFMAs", d, threads);
        run<0, 6, 0>("6 ds_read_b128 per 16 MFMAs", d, threads);
        run<8, 6, 0>("8 VALU + 6 ds_read_b128 per 16 MFMAs (main loop today)", d, threads);
        run<8, 6, 330>("phased: main loop + 330 VALU alone per pass (today)", d, threads);
        run<20, 6, 0>("interleaved: main loop + epilogue spread (20 VALU + 6 reads)", d, threads);
        run<26, 6, 0>("interleaved: 26 VALU + 6 reads per 16 MFMAs", d, threads);
    }
    return 0;
}
