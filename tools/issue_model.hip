// issue_model.hip — micro-benchmark of how MFMA, VALU and LDS reads of one or two waves per SIMD share a gfx950 SIMD
// (not part of the product; it decides how score_coarse.hip orders its main loop).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_model.hip -o tools/bin/issue_model && tools/bin/issue_model
// Every variant runs REPS x 28 int8 MFMAs (16x16x64, 16 cycles each at peak) per wave, plus the named extra work;
// one block per CU, 4 waves (one per SIMD) or 8 waves (two per SIMD). Reported: cycles per MFMA per SIMD
// (16.0 = matrix pipe saturated) from the event time at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define MFMA(c, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define VIND(x, y) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(msk))   // independent of the previous VALU
#define VDEP(x) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(msk))               // depends on the previous VALU on x
#define VMUL(x) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(x) : "v"(msk))
#define LDSR(b, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(addr))
#define WAITL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// MODE 0: MFMAs only                      1: per MFMA + 2 independent VALU        2: per MFMA + 2 dependent VALU (one chain)
//      3: per MFMA + 3 independent VALU   4: phases: 48 dependent VALU, then 28 MFMAs   5: phases: 48 independent VALU, then 28 MFMAs
//      6: phases: 7 LDS reads + wait, then 28 MFMAs        7: per 4 MFMAs one LDS read into the operand just used (in place)
//      8: phases: 7 LDS reads, 48 dependent VALU, wait, 28 MFMAs (the separated-phase main loop)
//      9: per MFMA 2 dependent VALU on TWO alternating chains + in-place LDS reads (the pipelined main loop)
//     10: VALU only: 48 dependent          11: VALU only: 48 independent          12: per MFMA + 2 dependent v_mul_u32_u24
template <int MODE, int PRIO>
__global__ void __launch_bounds__(512) k(int* out, int reps, uint32_t seed) {
    __shared__ i32x4 lds[64 * 8];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < 64u * 8u; i += blockDim.x) lds[i] = (i32x4){(int)i, 1, 2, 3};
    __syncthreads();
    i32x4 acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = (i32x4){0, 0, 0, 0};
    i32x4 A[4], B[7];
#pragma unroll
    for (int i = 0; i < 4; i++) A[i] = (i32x4){(int)(lane & 1u), 1, 0, 1};
#pragma unroll
    for (int i = 0; i < 7; i++) B[i] = (i32x4){1, (int)(lane & 3u), 0, 1};
    uint32_t msk = seed | 0x01010101u, x0 = lane + seed, x1 = lane * 3u, x2 = lane * 5u, x3 = lane * 7u, y0 = 1, y1 = 2, y2 = 3, y3 = 4;
    const uint32_t laddr = (uint32_t)(size_t)lds + lane * 16u;
    if (PRIO && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    if (MODE >= 13) {
        // 13: mode 8 with the step unrolled 8 times (4+ KB loop body); 14: + every 16 steps an "epilogue" of 700 VALU;
        // 15: as 14 with real operand expansion (bfe, mul_u24 with literal, and with literal) writing the A registers
        for (int r = 0; r < reps; r += 16) {
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
#pragma unroll
                    for (int t = 0; t < 7; t++) LDSR(B[t], laddr + t * 1024u + j * 16u);
                    if (MODE == 15) {
#pragma unroll
                        for (int rt = 0; rt < 4; rt++)
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                uint32_t tq;
                                asm volatile("v_bfe_u32 %0, %1, %2, 4" : "=v"(tq) : "v"(rt == 0 ? x0 : rt == 1 ? x1 : rt == 2 ? x2 : x3), "n"((j & 1) * 16 + q * 4));
                                asm volatile("v_mul_u32_u24 %0, 0x204081, %0" : "+v"(tq));
                                asm volatile("v_and_b32 %0, 0x1010101, %1" : "=v"(A[rt][q]) : "v"(tq));
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < 12; i++) { VDEP(x0); VDEP(x0); VDEP(x0); VDEP(x0); }
                    }
                    WAITL();
#pragma unroll
                    for (int t = 0; t < 7; t++)
#pragma unroll
                        for (int rt = 0; rt < 4; rt++) MFMA(acc[t * 4 + rt], A[rt], B[t]);
                }
            }
            if (MODE >= 14) {
#pragma unroll
                for (int i = 0; i < 175; i++) { VDEP(x0); VDEP(x1); VDEP(x2); VDEP(x3); }
            }
        }
    } else
    for (int r = 0; r < reps; r++) {
        if (MODE == 4 || MODE == 5 || MODE == 8 || MODE == 10 || MODE == 11) {
            if (MODE == 8) {
#pragma unroll
                for (int t = 0; t < 7; t++) LDSR(B[t], laddr + t * 1024u);
            }
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (MODE == 5 || MODE == 11) {
                    VIND(y0, x0); VIND(y1, x1); VIND(y2, x2); VIND(y3, x3);
                } else {
                    VDEP(x0); VDEP(x0); VDEP(x0); VDEP(x0);
                }
            }
            if (MODE == 8) WAITL();
        }
        if (MODE == 6) {
#pragma unroll
            for (int t = 0; t < 7; t++) LDSR(B[t], laddr + t * 1024u);
            WAITL();
        }
        if (MODE != 10 && MODE != 11) {
#pragma unroll
            for (int t = 0; t < 7; t++) {
#pragma unroll
                for (int rt = 0; rt < 4; rt++) {
                    MFMA(acc[t * 4 + rt], A[rt], B[t]);
                    if (MODE == 1) { VIND(y0, x0); VIND(y1, x1); }
                    if (MODE == 2) { VDEP(x0); VDEP(x0); }
                    if (MODE == 3) { VIND(y0, x0); VIND(y1, x1); VIND(y2, x2); }
                    if (MODE == 9) { if (rt & 1) { VDEP(x0); VDEP(x0); } else { VDEP(x1); VDEP(x1); } }
                    if (MODE == 12) { VMUL(x0); VMUL(x0); }
                }
                if (MODE == 7 || MODE == 9) LDSR(B[t], laddr + t * 1024u);
            }
            if (MODE == 7 || MODE == 9) WAITL();
        }
    }
    i32x4 s = (i32x4){(int)(x0 ^ x1 ^ x2 ^ x3 ^ y0 ^ y1 ^ y2 ^ y3), 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 28; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}


// Closer replicas of the coarse kernel's pass: 16 steps of [7 LDS reads from a 114 KB operand area | real expansion |
// 28 MFMAs] + a 700-VALU epilogue. FLAGS bit 0: global loads (8 B x 8 per 8 steps, consumed at once);
// bit 1: two-op expansion ((w >> j) & 0x01010101) instead of three-op; bit 2: no epilogue; bit 3: 512 MFMAs' worth
// of accumulators are zeroed with v_mov each pass.
template <int FLAGS>
__global__ void __launch_bounds__(512) k2(int* out, const uint2* rows, int reps, uint32_t seed) {
    extern __shared__ i32x4 big[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 2u * 8u * 7u * 64u; i += blockDim.x) {
        if (FLAGS & 128)  // realistic operand bytes (data-dependent power)
            big[i] = (i32x4){(int)(i * 2654435761u), (int)(i * 40503u + 77u) * 0x01000193, (int)((i ^ 0x5bd1e995u) * 2246822519u), (int)(i * 3266489917u + 1u)};
        else
            big[i] = (i32x4){(int)(i & 1u), 1, 0, 1};
    }
    __syncthreads();
    i32x4 acc[28];
#pragma unroll
    for (int i = 0; i < 28; i++) acc[i] = (i32x4){0, 0, 0, 0};
    i32x4 A[4], B[7];
#pragma unroll
    for (int i = 0; i < 4; i++) A[i] = (i32x4){(int)(lane & 1u), 1, 0, 1};
    uint32_t msk = seed | 0x01010101u, x0 = lane + seed, x1 = lane * 3u, x2 = lane * 5u, x3 = lane * 7u;
    uint32_t w[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) w[a][b] = lane * (a * 4 + b + 1) + seed;
    const uint32_t laddr = (uint32_t)(size_t)big + lane * 16u;
    const uint2* rp = rows + ((size_t)blockIdx.x * 8u + wave) * 64u * 17u * 2048u / 64u + lane * 17u;
    if ((FLAGS & 16) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_sleep(60);   // start waves 4-7 late
    if ((FLAGS & 32) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    if ((FLAGS & 64) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {  // waves 4-7 start with their epilogue
#pragma unroll
        for (int i = 0; i < 175; i++) { VDEP(x0); VDEP(x1); VDEP(x2); VDEP(x3); }
    }
    __shared__ int pass_counter;
    if (threadIdx.x == 0) pass_counter = 0;
    __syncthreads();
    const int total_passes = (reps / 16) * (int)(blockDim.x >> 6);
    if ((FLAGS & 512) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_barrier();  // waves 4-7 one phase behind
    for (int r = 0; r < reps || (FLAGS & 256); r += 16) {
        if (FLAGS & 256) {  // the block's passes are handed out dynamically: a wave that runs ahead simply takes more
            int t = 0;
            if (lane == 0) t = atomicAdd(&pass_counter, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= total_passes) break;
            r = (t >> 3) * 16;  // which rows: only the address pattern matters here
        }
        if (FLAGS & 8) {
#pragma unroll
            for (int i = 0; i < 28; i++) asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0" : "=v"(acc[i][0]), "=v"(acc[i][1]), "=v"(acc[i][2]), "=v"(acc[i][3]));
        }
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            if (FLAGS & 1) {
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        const uint2 v = rp[(size_t)(r / 16) * 64u * 17u + a * 16u * 17u + h * 8 + b];
                        w[a][2 * b] = v.x;
                        w[a][2 * b + 1] = v.y;
                    }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int t = 0; t < 7; t++) LDSR(B[t], laddr + (uint32_t)(h * 8 + j) * 7168u + t * 1024u);
#pragma unroll
                for (int rt = 0; rt < 4; rt++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        uint32_t tq;
                        if (FLAGS & 2) {
                            asm volatile("v_lshrrev_b32 %0, %2, %1" : "=v"(tq) : "v"(w[rt][q]), "n"(j));
                            asm volatile("v_and_b32 %0, 0x1010101, %1" : "=v"(A[rt][q]) : "v"(tq));
                        } else {
                            asm volatile("v_bfe_u32 %0, %1, %2, 4" : "=v"(tq) : "v"(w[rt][j >> 1]), "n"((j & 1) * 16 + q * 4));
                            asm volatile("v_mul_u32_u24 %0, 0x204081, %0" : "+v"(tq));
                            asm volatile("v_and_b32 %0, 0x1010101, %1" : "=v"(A[rt][q]) : "v"(tq));
                        }
                    }
                WAITL();
                if (FLAGS & 512) __builtin_amdgcn_s_barrier();  // explicit hand-off: one half of the waves multiplies
#pragma unroll
                for (int t = 0; t < 7; t++)
#pragma unroll
                    for (int rt = 0; rt < 4; rt++) MFMA(acc[t * 4 + rt], A[rt], B[t]);
                if (FLAGS & 512) __builtin_amdgcn_s_barrier();  // while the other half loads and expands
            }
        }
        if (!(FLAGS & 4)) {
#pragma unroll
            for (int i = 0; i < 175; i++) { VDEP(x0); VDEP(x1); VDEP(x2); VDEP(x3); }
        }
    }
    if ((FLAGS & 512) && __builtin_amdgcn_readfirstlane(threadIdx.x) < 256) __builtin_amdgcn_s_barrier();  // barrier counts match
    i32x4 s = (i32x4){(int)(x0 ^ x1 ^ x2 ^ x3), 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 28; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int FLAGS>
static void run2(const char* name, int* d, const uint2* rows) {
    const int reps = 2048;
    const size_t lds = 2 * 8 * 7 * 1024;
    hipFuncSetAttribute((const void*)k2<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<FLAGS>), dim3(256), dim3(512), lds, 0, d, rows, reps, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<FLAGS>), dim3(256), dim3(512), lds, 0, d, rows, reps, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s 2 waves/SIMD: %7.3f ms  %6.2f cycles per MFMA slot per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)reps * 28 * 2));
}

template <int MODE, int PRIO>
static void run(const char* name, int* d) {
    const int reps = 2048;
    for (int waves : {4, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(waves * 64), 0, 0, d, reps, 12345u);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, PRIO>), dim3(256), dim3(waves * 64), 0, 0, d, reps, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double units = (double)reps * 28 * (waves / 4);  // MFMAs (or 28-op groups) per SIMD
        printf("%-58s %d waves/SIMD: %7.3f ms  %6.2f cycles per MFMA slot per SIMD\n", name, waves / 4, ms, ms * 1e-3 * 2.4e9 / units);
    }
}

int main() {
    int* d;
    hipMalloc(&d, 256 * 512 * 4);
    run<0, 0>("0 MFMA only", d);
    run<1, 0>("1 MFMA + 2 independent VALU each", d);
    run<2, 0>("2 MFMA + 2 dependent VALU each", d);
    run<12, 0>("12 MFMA + 2 dependent v_mul_u32_u24 each", d);
    run<3, 0>("3 MFMA + 3 independent VALU each", d);
    run<9, 0>("9 MFMA + 2 dep VALU (2 chains) + in-place LDS reads", d);
    run<7, 0>("7 MFMA + in-place LDS read per 4", d);
    run<4, 0>("4 phases: 48 dependent VALU | 28 MFMA", d);
    run<5, 0>("5 phases: 48 independent VALU | 28 MFMA", d);
    run<6, 0>("6 phases: 7 LDS reads + wait | 28 MFMA", d);
    run<8, 0>("8 phases: 7 LDS reads, 48 dep VALU, wait | 28 MFMA", d);
    run<8, 1>("8 same, waves 4-7 at s_setprio 1", d);
    run<4, 1>("4 phases 48 dep VALU | 28 MFMA, waves 4-7 at s_setprio 1", d);
    run<13, 0>("13 mode 8, step unrolled x8 (big loop body)", d);
    run<14, 0>("14 = 13 + 700-VALU epilogue per 16 steps", d);
    run<15, 0>("15 = 14 with the real expansion writing A", d);
    run<15, 1>("15 with waves 4-7 at s_setprio 1", d);
    uint2* rows;
    hipMalloc(&rows, (size_t)256 * 8 * 128 * 64 * 17 * 8 + (1 << 20));
    hipMemset(rows, 0x5a, (size_t)256 * 8 * 128 * 64 * 17 * 8 + (1 << 20));
    run2<4>("k2: 114 KB LDS operands, 3-op expansion, no epilogue", d, rows);
    run2<0>("k2: + 700-VALU epilogue", d, rows);
    run2<1>("k2: + global row loads", d, rows);
    run2<3>("k2: + global row loads, 2-op expansion", d, rows);
    run2<11>("k2: + loads, 2-op expansion, v_mov zeroing", d, rows);
    run2<7>("k2: loads, 2-op expansion, no epilogue", d, rows);
    run2<6>("k2: 2-op expansion, no loads, no epilogue", d, rows);
    run2<7 + 128>("k2: loads, 2-op expansion, no epilogue, RANDOM operand bytes", d, rows);
    run2<3 + 128>("k2: loads, 2-op expansion, epilogue, RANDOM operand bytes", d, rows);
    run2<6 + 512>("k2: 2-op, no loads, no epilogue, barrier hand-off between wave halves", d, rows);
    run2<7 + 512>("k2: loads, 2-op, no epilogue, barrier hand-off", d, rows);
    run2<3 + 512>("k2: loads, 2-op, epilogue, barrier hand-off", d, rows);
    run2<3 + 256>("k2: loads, 2-op, epilogue, dynamic passes", d, rows);
    run2<3 + 256 + 32>("k2: loads, 2-op, epilogue, dynamic passes + waves 4-7 s_setprio 1", d, rows);
    run2<7 + 256>("k2: loads, 2-op, no epilogue, dynamic passes", d, rows);
    run2<7 + 256 + 32>("k2: loads, 2-op, no epilogue, dynamic passes + waves 4-7 s_setprio 1", d, rows);
    run2<3 + 16>("k2: loads, 2-op, epilogue, waves 4-7 start late", d, rows);
    run2<3 + 32>("k2: loads, 2-op, epilogue, waves 4-7 s_setprio 1", d, rows);
    run2<3 + 16 + 32>("k2: loads, 2-op, epilogue, late + prio", d, rows);
    run2<3 + 64>("k2: loads, 2-op, epilogue, waves 4-7 epilogue first", d, rows);
    run<10, 0>("10 VALU only: 48 dependent (per 28-slot group)", d);
    run<11, 0>("11 VALU only: 48 independent (per 28-slot group)", d);
    return 0;
}
