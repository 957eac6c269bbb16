// probe_mx_pingpong.hip — does v_mfma_scale_f32_16x16x128_f8f6f4 run slower when its result goes to OTHER registers than
// its C operand came from (diagnostics, not product)? The compiler's register allocation of score_mx.hip rotates the
// accumulators through the registers of dying B operands: slice 0 writes D = tmp <- C = acc, slice 1 four instructions
// later writes D = acc' <- C = tmp. In-place accumulation (D = C) is what tools/probe_mx_issue.hip measured.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mx_pingpong.hip -o tools/bin/probe_mx_pingpong
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define M6(d, a, b, c, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "=v"(d) : "v"(a), "v"(b), "v"(c), "v"(sa), "v"(sb))
#define M4(d, a, b, c, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %3, %4, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "=v"(d) : "v"(a), "v"(b), "v"(c), "v"(sa), "v"(sb))
#define M6I(c, a, b, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define M4I(c, a, b, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb))

// MODE 0: in place, the order of the kernel (4 row tiles x slice 0, then 4 x slice 1, per column tile)
// MODE 1: slice 0 into four temporaries, slice 1 from the temporaries back into the accumulators
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63u;
    v4i A[4], B4;
    v6i B6;
#pragma unroll
    for (int i = 0; i < 4; i++) A[i] = (v4i){(int)(0x11111111u & (lane * 0x01010101u + seed)), 0x22222222 & (int)seed, 0x11110000, 0x00001111};
    B6 = (v6i){0x08208208, 0x20820820, (int)0x82082082u, 0x08208208, 0x20820820, (int)0x82082082u};
    B4 = (v4i){0x12341234, 0x21212121, 0x11111111, 0x22222222};
    const int sa = 0x7F7F7F7F, sb = 0x84848484;
    v4f acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = (v4f){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 7; t++) {
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) M6I(acc[t * 4 + r], A[r], B6, sa, sb);
#pragma unroll
                for (int r = 0; r < 4; r++) M4I(acc[t * 4 + r], A[r], B4, sa, sa);
            } else {
                v4f tmp[4];
#pragma unroll
                for (int r = 0; r < 4; r++) M6(tmp[r], A[r], B6, acc[t * 4 + r], sa, sb);
#pragma unroll
                for (int r = 0; r < 4; r++) M4(acc[t * 4 + r], A[r], B4, tmp[r], sa, sa);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 28; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* d, int threads) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, iters, 0x01010101u + rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-70s %d wave(s)/SIMD: %8.3f ms  %5.2f ns per MFMA and SIMD\n", name, threads / 256, best, best * 1e6 / ((double)iters * 56.0 * (threads / 256)));
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<0>("in place (D = C)", d, threads);
        run<1>("slice 0 into temporaries, slice 1 back (D != C, distance 4)", d, threads);
    }
    return 0;
}
