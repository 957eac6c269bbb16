// score_mx32.hip — the block-scaled coarse filter of score_mx.hip on the LARGE shape of the instruction,
// v_mfma_scale_f32_32x32x64_f8f6f4: the same FP4 table bits x FP6 + FP4 phenotype slices, the same rigorous bound, survivors'
// bitmap and exact re-scoring behind it (calculate_kmer_score, src/kmers_multiple_databases.cpp:327-363) - half as many matrix
// instructions for the same multiply-adds. A 32 x 32 x 64 instruction occupies the pipe twice as long as a 16 x 16 x 128 one
// and hides about four other vector instructions of its SIMD where the small shape hides one (tools/probe_mx32.hip); the
// filter issues ~1.7 per small-shape instruction, i.e. ~3.5 per large one.
//
//  * a wave pass is 64 table rows = TWO row tiles of 32; the column tiles are CT32 tiles of 32 columns (two slices each:
//    an FP6 and an FP4 instruction into the same accumulator, as in score_mx.hip) and, for a remainder of up to 16 columns,
//    ONE "combined" tile: slice 0 of its 16 columns in B-operand lanes 0..15, slice 1 in lanes 16..31, both as FP6 codes
//    (every E2M1 value is an E2M3 value) with PER-LANE block scales (2^5 / 2^0) - one instruction does both slices of 16
//    columns, the two partial sums meet in the epilogue (one cross-lane add per row). 102 operand columns (101 + ones) are
//    3 x 32 + 16 = 112 executed columns, as with seven tiles of 16.
//  * table row permutation: MFMA row i of a row tile is table row rho(i) = (i2 << 4 | i4 << 3 | i3 << 2 | i1 << 1 | i0), so
//    that a lane's 16 accumulator registers of a tile are the 16 CONSECUTIVE table rows 16 h + r (h = lane >> 5): a column's
//    survivors of a pass leave as plain 16-bit quarters of its 64-row bitmap word (no transposition to undo), and the row
//    terms a lane needs are 16 consecutive entries of the exchange area.
//  * samples: groups of 256 (32 bytes of a row: the two k-lanes of a row fetch them as one contiguous piece, 16 bytes each),
//    four K = 64 steps per group (step j takes bit 4 e' + j of dword q as k = 32 kblk + 8 q + e'); the rest in quarter steps
//    of 64 samples (the lane's own dword shifted by 0..3).
//
// Operand and result maps of the instruction: tools/probe_mx32map.hip (checked on the device, 0 mismatches).
//   A: lane (i = lane & 31, kblk = lane >> 5): k = 32 kblk + e in nibble e          B: lane (j = lane & 31, kblk): k = 32 kblk + e in field e
//   D: lane (j, h = lane >> 5), register r: row (r & 3) + 8 (r >> 2) + 4 h, column j
#include <stdlib.h>

#include <algorithm>

#include "score_common.h"

// Timing experiments only (wrong results). Bits: 1 the tests are replaced by an XOR over all accumulators (every MFMA stays
// alive), 2 no cross-lane add for the combined tile, 4 accumulators not zeroed per pass, 8 no row loads.
#ifndef KGWAS_MX32_ABLATE
#define KGWAS_MX32_ABLATE 0
#endif

namespace kgwas {

typedef int m32v8i __attribute__((ext_vector_type(8)));
typedef float m32v16f __attribute__((ext_vector_type(16)));

namespace {
constexpr uint32_t M32_FULL = 2560u;  // a full tile of one K = 64 step: 64 lanes x 16 B | x 8 B (FP6) | x 16 B (FP4)
constexpr uint32_t M32_COMB = 1536u;  // the combined tile: FP6 only
}

__host__ __device__ constexpr uint32_t mx32_step_bytes(int ct32, int comb) { return (uint32_t)ct32 * M32_FULL + (uint32_t)comb * M32_COMB; }

template <int CT32, int COMB, int TH>
__global__ void __launch_bounds__(TH) mx32_kernel(MxArgs a, uint32_t rows_per_block, uint32_t n_rowblocks, uint32_t grid_lg) {
    extern __shared__ uint4 mlds32[];  // [n_steps][step bytes], colc[3][NT*32] (alpha, -, column index), per-wave row-term exchange
    constexpr int NT = CT32 + COMB;     // tiles (units of the pipeline)
    constexpr uint32_t SB = mx32_step_bytes(CT32, COMB);
    constexpr int SLOTS = NT * 32;
    uint32_t rb = blockIdx.x, lg0 = 0, lg1 = a.n_lgroups;
    if (grid_lg) {  // every (row block, LDS group) pair is a block; the groups of a row block run next to each other on one XCD
        const uint32_t idx = blockIdx.x >> 3;
        rb = (idx / a.n_lgroups) * 8u + (blockIdx.x & 7u);
        lg0 = idx % a.n_lgroups;
        lg1 = lg0 + 1u;
    }
    if (rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t h = lane >> 5, j32 = lane & 31u;
    // table row of this lane's A operand inside a row tile (see the header): rho(i)
    const uint32_t rho = ((j32 >> 2) & 1u) << 4 | ((j32 >> 4) & 1u) << 3 | ((j32 >> 3) & 1u) << 2 | (j32 & 3u);
    const uint32_t n_steps = 4u * a.n_full + a.n_quarter;
    const uint32_t group_bytes = n_steps * SB;
    char* lds = reinterpret_cast<char*>(mlds32);
    float* colc = reinterpret_cast<float*>(lds + group_bytes);
    const int* colp = reinterpret_cast<const int*>(colc + 2 * SLOTS);
    float* wscr = colc + 3 * SLOTS + wave * 192u;  // wave-private: 64 x N1, 64 x (sqrt(d), E)
    const uint32_t rows_per_pass = (TH / 64) * 64u;
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    const int sc0 = (int)a.scale0;  // block scale of the first slice
    // the combined tile's per-lane block scale: first slice in lanes 0..15, second (2^0) in lanes 16..31
    const int sc_comb = j32 < 16u ? sc0 : 0x7F7F7F7F;
    uint32_t tested_local = 0;

    for (uint32_t lg = lg0; lg < lg1; lg++) {
        if (lg != lg0) __syncthreads();
        {
            const uint4* src = reinterpret_cast<const uint4*>(a.Bq + (size_t)lg * group_bytes);
            for (uint32_t i = threadIdx.x; i < group_bytes / 16u; i += TH) mlds32[i] = src[i];
            if (threadIdx.x < SLOTS) {
                const CoarseCol cc = a.cols[lg * SLOTS + threadIdx.x];
                float al = __builtin_huge_valf();  // padding / ones column / the combined tile's second half: nothing survives
                if (cc.pheno >= 0) al = (float)(sqrt(a.thr[cc.pheno]) * cc.kalpha);  // NaN threshold (frozen column) -> NaN -> nothing survives
                colc[threadIdx.x] = al;
                colc[SLOTS + threadIdx.x] = cc.iu;
                reinterpret_cast<int*>(colc + 2 * SLOTS)[threadIdx.x] = cc.pheno;
            }
        }
        __syncthreads();

        uint32_t ro[2];  // 32-bit byte offsets of this lane's two rows (launch_mx32 guarantees the chunk spans < 4 GiB)
        const uint64_t wave_row0 = blk_row0 + wave * 64u;
        auto set_rows = [&](uint32_t (&o)[2], uint64_t rb0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                uint64_t r = rb0 + rt * 32u + rho;
                if (r >= a.n_rows) r = a.n_rows - 1;
                o[rt] = ((uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u;
            }
        };
        struct __attribute__((aligned(8))) U4 {
            uint32_t x, y, z, w;
        };
        // this lane's 16 bytes of 256-sample group g of its two rows
        auto load_group = [&](uint32_t (&pc)[2][4], const uint32_t (&o)[2], uint32_t g) {
            const uint32_t b0 = 32u * g + 16u * h;
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                uint32_t off = o[rt] + b0;
                asm volatile("" : "+v"(off));  // a 32-bit offset on the scalar base, made here: not a hoisted (and spilled) 64-bit pointer
                const U4 v = (KGWAS_MX32_ABLATE & 8) ? U4{off, lane * 2654435761u, lane, b0} : *reinterpret_cast<const U4*>(rows_base + off);
                pc[rt][0] = v.x;
                pc[rt][1] = v.y;
                pc[rt][2] = v.z;
                pc[rt][3] = v.w;
            }
        };
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        // B operands, read one UNIT ahead of their MFMAs. A unit is one slice of one tile (two MFMAs: the two row tiles); a step
        // runs the first slices of all its tiles, then the second slices: the two instructions that add into the same
        // accumulator are a whole phase (2 NT instructions) apart, not two (with the second slice right behind the first -
        // a dependent 16-pass instruction two issue slots later - the main loop measured 13 % slower than the 16 x 16 x 128
        // kernel's). Unit 0 of a step has a register set of its own (it is fetched while the LAST unit of the step before
        // multiplies), the others alternate between two more.
        constexpr int NU = NT + CT32;
        u32x4 Bx[3];
        u32x2 By[3];
        auto set_of = [](int u) { return u == 0 ? 2 : ((u - 1) & 1); };
        struct StepAddr {
            uint32_t o16, o8;
        };
        auto step_addr = [&](const char* bstep) {
            StepAddr sa;
            sa.o16 = (uint32_t)(bstep - lds) + lane * 16u;
            sa.o8 = (uint32_t)(bstep - lds) + lane * 8u + 1024u;
            asm volatile("" : "+v"(sa.o16), "+v"(sa.o8));
            return sa;
        };
        auto read_unit = [&](int u, int set, const StepAddr& sa) {
            if (u < NT) {  // first slice of tile u: FP6, dwords 0-3 | 4-5
                Bx[set] = *reinterpret_cast<const u32x4*>(lds + sa.o16 + u * M32_FULL);
                By[set] = *reinterpret_cast<const u32x2*>(lds + sa.o8 + u * M32_FULL);
            } else {  // second slice of tile u - NT: FP4, dwords 0-3
                Bx[set] = *reinterpret_cast<const u32x4*>(lds + sa.o16 + (u - NT) * M32_FULL + 1536);
            }
        };
        uint32_t piece[2][4];
        set_rows(ro, wave_row0);
        if (a.n_full && wave_row0 < a.n_rows) load_group(piece, ro, 0);
        StepAddr sadr = step_addr(lds);
        read_unit(0, 2, sadr);  // step 0 of the first pass; every pass's last step fetches it for the next
        for (uint32_t ps = 0; ps * rows_per_pass < rows_per_block; ps++) {
            const uint64_t rbase = wave_row0 + (uint64_t)ps * rows_per_pass;
            if (rbase >= a.n_rows) break;  // wave-uniform
            uint32_t ro_next[2];
            set_rows(ro_next, rbase + rows_per_pass);
            m32v16f acc[2][NT];
            if (!(KGWAS_MX32_ABLATE & 4) || ps == 0) {
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[rt][t][r] = 0.0f;
            }

            // one step: [read unit 1 | MFMAs of unit 0 | read unit 2 | MFMAs of unit 1 | ... | read the next step's unit 0 | MFMAs of the last unit]
            auto run_step = [&](const m32v8i (&A)[2], const char* bs_next, int sa) {
                const StepAddr nadr = step_addr(bs_next);
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int set = set_of(u);
                    if (u + 1 < NU)
                        read_unit(u + 1, set_of(u + 1), sadr);
                    else
                        read_unit(0, NU == 1 ? 0 : 2, nadr);  // (a single unit per step: its own set is still being multiplied with - moved over below)
                    __builtin_amdgcn_sched_barrier(0);
                    if (u < NT) {
                        const m32v8i B0 = {(int)Bx[set].x, (int)Bx[set].y, (int)Bx[set].z, (int)Bx[set].w, (int)By[set].x, (int)By[set].y, 0, 0};
                        const int sb = (COMB && u == NT - 1) ? sc_comb : sc0;
#pragma unroll
                        for (int rt = 0; rt < 2; rt++) acc[rt][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[rt], B0, acc[rt][u], 4, 2, 0, sa, 0, sb);
                    } else {
                        const m32v8i B1 = {(int)Bx[set].x, (int)Bx[set].y, (int)Bx[set].z, (int)Bx[set].w, 0, 0, 0, 0};
#pragma unroll
                        for (int rt = 0; rt < 2; rt++)
                            acc[rt][u - NT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[rt], B1, acc[rt][u - NT], 4, 4, 0, sa, 0, 0x7F7F7F7F);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (NU == 1) Bx[2] = Bx[0], By[2] = By[0];
                sadr = nadr;
            };

            __builtin_amdgcn_s_setprio(0);
            for (uint32_t g = 0; g < a.n_full; g++) {
                const char* bg = lds + (size_t)g * 4u * SB;
                const bool last_g = g + 1u == a.n_full;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    m32v8i A[2];
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) {
                        A[rt] = (m32v8i){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            A[rt][q] = j < 3 ? (int)(piece[rt][q] & (0x11111111u << j)) : (int)((piece[rt][q] >> 1) & 0x44444444u);
                    }
                    const char* bs_next = bg + (j + 1) * SB;
                    if (j == 3) {
                        // (selects, not branches: see score_mx.hip) the NEXT group's pieces - or group 0 of the wave's next rows
                        __builtin_amdgcn_sched_barrier(0);
                        uint32_t on[2];
#pragma unroll
                        for (int rt = 0; rt < 2; rt++) on[rt] = last_g ? ro_next[rt] : ro[rt];
                        load_group(piece, on, last_g ? 0u : g + 1u);
                        if (last_g && a.n_quarter == 0) bs_next = lds;  // the pass's last step: step 0 of the next pass
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    run_step(A, bs_next, j == 0 ? 0x7F7F7F7F : j == 1 ? 0x7E7E7E7E : 0x7D7D7D7D);
                }
            }
            // quarter steps: 64 samples each, the lane's own dword shifted by 0..3
            for (uint32_t x = 0; x < a.n_quarter; x++) {
                uint32_t b0 = 32u * a.n_full + 8u * x + 4u * h;
                b0 = b0 + 4u <= avail_b ? b0 : avail_b - 4u;  // (a clamped lane's elements meet zeros in every B operand)
                m32v8i A[2];
#pragma unroll
                for (int rt = 0; rt < 2; rt++) {
                    const uint32_t w = *reinterpret_cast<const uint32_t*>(rows_base + (ro[rt] + b0));
                    A[rt] = (m32v8i){(int)(w & 0x11111111u), (int)((w >> 1) & 0x11111111u), (int)((w >> 2) & 0x11111111u),
                                     (int)((w >> 3) & 0x11111111u), 0, 0, 0, 0};
                }
                const char* bs = lds + (size_t)(4u * a.n_full + x) * SB;
                run_step(A, x + 1u == a.n_quarter ? lds : bs + SB, 0x7F7F7F7F);
            }
            __builtin_amdgcn_s_setprio(3);  // the epilogue issues vector instructions only: through them quickly (score_mx.hip)

            // ---- epilogue -----------------------------------------------------------------------------------------
            // the combined tile: a column's two partial sums sit 16 lanes apart
            if (COMB && !(KGWAS_MX32_ABLATE & 2)) {
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float o = __shfl_xor(acc[rt][NT - 1][r], 16);
                        acc[rt][NT - 1][r] = acc[rt][NT - 1][r] + o;  // (exact: multiples of kappa far below 2^24 kappa; both halves now hold the sum)
                    }
            }
            // N1 of the pass's 64 rows: the ones column is slot 31 of the last tile (lane 31 of each half holds rows 16 h + r)
            float2* trm = reinterpret_cast<float2*>(wscr + 64);
            {
                float* n1s = wscr;
                if (j32 == 31u) {
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
#pragma unroll
                        for (int r = 0; r < 16; r += 4)
                            *reinterpret_cast<float4*>(n1s + rt * 32 + 16 * h + r) =
                                make_float4(acc[rt][NT - 1][r], acc[rt][NT - 1][r + 1], acc[rt][NT - 1][r + 2], acc[rt][NT - 1][r + 3]);
                }
                __builtin_amdgcn_wave_barrier();
                const uint64_t left = a.n_rows - rbase;
                const uint32_t rows_here = left < 64u ? (uint32_t)left : 64u;
                const bool mac_any = a.S >= 2u * a.min_count;
                const uint32_t span = a.S - 2u * a.min_count;
                const float f = n1s[lane];  // row `lane` of the pass
                const uint32_t n1r = (uint32_t)f;
                const bool ok = mac_any & (lane < rows_here) & ((n1r - a.min_count) <= span);
                if (lg == 0) tested_local += ok ? 1u : 0u;
                const float sq = __builtin_amdgcn_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; 1 ulp sqrt; (1 - 2^-20)
                float2 tm;
                tm.x = ok ? sq : __builtin_huge_valf();
                tm.y = (a.eg_max + fminf(a.rall_max, f * a.rmax_max)) * 1.000001f;
                trm[lane] = tm;
                __builtin_amdgcn_wave_barrier();
            }
            float alc[NT];
            float al_min = __builtin_huge_valf();
#pragma unroll
            for (int t = 0; t < NT; t++) {
                alc[t] = colc[t * 32 + j32];
                al_min = fminf(al_min, alc[t]);  // (NaN: frozen column, skipped; +inf: padding / ones / second half of the combined tile)
            }
            // lanes whose accumulator in the LAST tile is no margin: the ones column (N1) - slot 31 of a full tile, slot 15 of the
            // combined one - and the combined tile's upper half (the same sums as lanes 0..15, no columns of their own)
            const bool no_margin = COMB ? j32 >= 15u : j32 == 31u;
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                float sqd[16], er[16];
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float4 v = *reinterpret_cast<const float4*>(trm + rt * 32 + 16 * h + r);
                    sqd[r] = v.x;
                    er[r] = v.y;
                    sqd[r + 1] = v.z;
                    er[r + 1] = v.w;
                }
                float mx[16];
#pragma unroll
                for (int r = 0; r < 16; r++) mx[r] = no_margin ? 0.0f : fabsf(acc[rt][NT - 1][r]);
#pragma unroll
                for (int t = 0; t + 1 < NT; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) mx[r] = fmaxf(mx[r], fabsf(acc[rt][t][r]));
                uint64_t hit[16];
                uint64_t hit_any = 0;
                if (KGWAS_MX32_ABLATE & 1) {
                    int x = 0;
#pragma unroll
                    for (int r = 0; r < 16; r++)
#pragma unroll
                        for (int t = 0; t < NT; t++) x ^= __float_as_int(acc[rt][t][r]);
#pragma unroll
                    for (int r = 0; r < 16; r++) hit[r] = 0;
                    hit[0] = __ballot(x == 0x7fffffff);
                    hit_any = hit[0];
                } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    hit[r] = __ballot(fmaf(-al_min, sqd[r], mx[r]) + er[r] >= 0.0f);
                    hit_any |= hit[r];
                }
                }
                if (hit_any) {  // wave-uniform
                    uint32_t mb[NT];
#pragma unroll
                    for (int t = 0; t < NT; t++) mb[t] = 0;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (hit[r]) {  // wave-uniform
#pragma unroll
                            for (int t = 0; t < NT; t++) {
                                float al = alc[t];
                                asm volatile("" : "+v"(al));
                                mb[t] |= (fmaf(-al, sqd[r], fabsf(acc[rt][t][r])) + er[r] >= 0.0f) ? (1u << r) : 0u;  // NaN never passes
                            }
                        }
                    }
                    // lane (j, h): rows 32 rt + 16 h + r of the pass = quarter 2 rt + h of the column's 64-row word, in row order
                    unsigned short* bm16 = reinterpret_cast<unsigned short*>(a.bitmap) + (rbase >> 6) * 4u + 2u * rt + h;
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        if (mb[t]) bm16[(uint64_t)colp[t * 32 + j32] * a.words_per_col * 4u] = (unsigned short)mb[t];  // column >= 0 wherever a bit is set
                }
            }
            __builtin_amdgcn_wave_barrier();  // the exchange area is rewritten by the next pass
            ro[0] = ro_next[0];
            ro[1] = ro_next[1];
        }
    }
    if (a.tested) {
        uint32_t v = tested_local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

size_t mx32_lds_bytes(uint32_t n_steps, uint32_t ct32, uint32_t comb) {
    return (size_t)n_steps * mx32_step_bytes((int)ct32, (int)comb) + 3u * (ct32 + comb) * 32u * 4u + 8u * 768u;
}

template <int CT32, int COMB>
static hipError_t launch_mx32_t(const MxArgs& a, uint32_t rows_per_block, size_t lds, hipStream_t st) {
    constexpr int TH = 512;
    const uint32_t rpp = (TH / 64) * 64u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mx32_kernel<CT32, COMB, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t grid_lg = a.n_lgroups > 1 ? 1u : 0u;
    const uint32_t grid = grid_lg ? (n_rowblocks + 7u) / 8u * 8u * a.n_lgroups : n_rowblocks;
    hipLaunchKernelGGL((mx32_kernel<CT32, COMB, TH>), dim3(grid), dim3(TH), lds, st, a, rows_per_block, n_rowblocks, grid_lg);
    return hipGetLastError();
}

// a: n_full = whole 256-sample groups, n_quarter = 64-sample steps behind them; Bq / cols in the layout of this file
hipError_t launch_mx32(const MxArgs& a, uint32_t ct32, uint32_t comb, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    if (a.n_slices != 2 || a.s1_fp6) return hipErrorInvalidValue;
    const size_t lds = mx32_lds_bytes(4u * a.n_full + a.n_quarter, ct32, comb);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if ((a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit byte offsets
    switch (ct32 * 2u + comb) {
        case 1: return launch_mx32_t<0, 1>(a, rows_per_block, lds, st);
        case 2: return launch_mx32_t<1, 0>(a, rows_per_block, lds, st);
        case 3: return launch_mx32_t<1, 1>(a, rows_per_block, lds, st);
        case 4: return launch_mx32_t<2, 0>(a, rows_per_block, lds, st);
        case 5: return launch_mx32_t<2, 1>(a, rows_per_block, lds, st);
        case 6: return launch_mx32_t<3, 0>(a, rows_per_block, lds, st);
        case 7: return launch_mx32_t<3, 1>(a, rows_per_block, lds, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace kgwas
