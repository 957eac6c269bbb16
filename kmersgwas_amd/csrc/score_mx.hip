// score_mx.hip — the coarse filter of the sparse phase on gfx950's block-scaled matrix instruction
// (v_mfma_scale_f32_16x16x128_f8f6f4): FP4 table bits x FP6 / FP4 phenotype slices, both slices of a column accumulated
// into ONE float32 accumulator through the block scales. Same contract as the int8 filter of score_coarse.hip (which it
// replaces by default): for EVERY (k-mer, column) pair an approximation of yigi = sum_i g_i*y_i of
// calculate_kmer_score (src/kmers_multiple_databases.cpp:327-363) with a RIGOROUS bound, survivors as bits of the
// chunk's [column][row] bitmap, exact re-scoring of the survivors by rescore_kernel - results bit-identical to the
// exact scorers'.
//
//  * table bits as FP4 (E2M1). A nibble with only bit 0, 1 or 2 set is 0.5, 1.0 or 2.0, so `dword & (0x11111111 << j)`
//    IS an operand of eight table bits for j = 0, 1, 2 (bit 3, the FP4 sign, is shifted down first), and the operand's
//    block scale 2^0 / 2^-1 / 2^-2 (exact) makes every set bit 0.5: five lane-ops per 32 table bits (int8: 16), and one
//    K = 128 instruction does the work of two K = 64 int8 ones in the same 16 pipe cycles.
//  * phenotype slices as FP6 (E2M3) + FP4 (or FP6 + FP6), NOT uniformly quantised: with the integer grids
//        A6 = {0..15, 16..30 step 2, 32..60 step 4}   (E2M3 values x 8)      A4 = {0, 1, 2, 3, 4, 6, 8, 12}  (E2M1 x 2)
//    a column is  y_i - c ~ w * t_i,  t_i = 2^s * a6_i + a_i  (s = 3 with an FP4 second slice, 5 with an FP6 one),
//    c = sum / N. The second slice's block scale is 2^0 and the first one's 2^5, so that BOTH products land in the same
//    accumulator in the same unit: acc = kappa * sum_i g_i t_i, kappa = 1/4 (FP4 second slice) or 1/16 - every partial
//    sum a multiple of kappa far below 2^24 * kappa, i.e. exact in float32 in any order. One accumulator per (row tile,
//    16 columns) instead of one per slice, one fma per pair in the test instead of a combine + fma, and a residual of
//    at most half a unit near zero (where most of a phenotype's values are), a unit up to twice that, two units at the
//    extremes: 2.5x tighter than the uniform 961-level grid the same two operands would give.
//  * the operands (the B side: 1.25 bytes per sample and column with an FP4 second slice, 2560 bytes per MFMA step and
//    16 columns) sit in LDS for the whole block, as before; a wave owns RT x 16 rows and CT column tiles' accumulators
//    (RT = 4; eight waves per block with 4..7 column tiles, twelve - three per SIMD, 168 registers - with up to three:
//    the 2048-sample shapes, whose operands only fit three tiles at a time).
//  * samples beyond the last full 512-sample group are taken in QUARTER groups of 128 (one MFMA step each, the lane's own
//    dword shifted by 0..3; up to four of them) instead of a whole padded group: 1135 samples are 9 steps, not 12.
//
// Sample <-> k maps (tools/probe_mx6.hip checks the FP6 operand's element order and the exactness of E2M3 subnormals
// and of the two-scale accumulation on the device):
//   full group g, step j:   lane (m = lane & 15, kb = lane >> 4) holds bytes 64 g + 16 kb .. +15 of row m's bits as
//                           dwords q = 0..3; nibble e' of operand dword q = bit 4 e' + j of dword q, k = 32 kb + 8 q + e'
//                           <-> sample 512 g + 128 kb + 32 q + 4 e' + j
//   quarter step x:         the lane holds dword kb of the 16 bytes at 64 G + 16 x; operand dword q = (dword >> q) &
//                           0x11111111, k = 32 kb + 8 q + e' <-> sample 512 G + 128 x + 32 kb + 4 e' + q
//   slice operands:         lane (n = lane & 15, kb) holds column n's values for k = 32 kb + e, e = 0..31, in 6-bit
//                           (FP6: 6 dwords) or 4-bit (FP4: 4 dwords) fields, little-endian
//   accumulators:           lane (n, kb): rows 4 kb + i of the row tile, column n
#include <stdlib.h>

#include <algorithm>

#include "score_common.h"

// Timing experiments only (wrong results). Bits: 1 the tests are replaced by an XOR over all accumulators (every MFMA
// stays alive), 2 no operand expansion, 4 no LDS operand reads, 8 no row loads, 32 no survivor emission.
#ifndef KGWAS_MX_ABLATE
#define KGWAS_MX_ABLATE 0
#endif

// KGWAS_MX_NT=1: non-temporal row loads, as in the narrow filter - measured 10.00 -> 10.25 ms per 100 M rows x 101 columns here
// (this kernel waits for its matrix pipe, not for the memory system; with several LDS groups the rows are meant to stay in L2).
#ifndef KGWAS_MX_NT
#define KGWAS_MX_NT 0
#endif

// KGWAS_MX_WARM=1 (experiments, measured SLOWER: 10.2 against 9.55 ms per 100 M rows x 1024 x 101): every wave asks for the cache
// lines of its NEXT pass's rows at the top of a pass - one `global_load_lds_dword` per 8 KB, a lane per 128-byte line, landing in
// a junk LDS word - so that the pass's real row loads (one step ahead of their use: all the registers allow) find the rows in
// the L2. The row loads' 1.5 ms in the ablations is not HBM latency waited out: under the board's power limit
// every extra request costs clock.
#ifndef KGWAS_MX_WARM
#define KGWAS_MX_WARM 0
#endif
// KGWAS_MX_PERSIST=1 (shapes with ONE LDS group; the default since round 6): as many blocks as the chip holds at once (one per
// CU: the operands fill the LDS), each taking the 512-row passes b, b + grid, b + 2 grid, ... of the chunk - the operands are
// copied into the LDS once per block and LAUNCH instead of once per 4096 rows, every CU gets the same number of passes, and no
// CU waits for a last long block while others are done: 9.0 against 9.5 ms per 100 M rows x 1024 x 101 in alternating runs
// (tools/mx_ab.sh). 0: one block per 512 .. 4096 rows, as in rounds 3-5.
#ifndef KGWAS_MX_PERSIST
#define KGWAS_MX_PERSIST 1
#endif

namespace kgwas {

typedef int mxv8i __attribute__((ext_vector_type(8)));
typedef float mxv4f __attribute__((ext_vector_type(4)));

namespace {
constexpr uint32_t MX_PART0 = 1536u;  // FP6 slice of one (step, column tile): 64 lanes x 16 B, then 64 lanes x 8 B
}

// S1F: format of the second slice (4 = FP4 E2M1, 2 = FP6 E2M3); NS = 1: first slice only.
#ifndef KGWAS_MX_PRIO
#define KGWAS_MX_PRIO 0
#endif
#ifndef KGWAS_MX_PRIO_EPI
#define KGWAS_MX_PRIO_EPI 3  // the epilogue runs at raised wave priority (see there); 0: off
#endif

template <int NS, int S1F>
__host__ __device__ constexpr uint32_t mx_step_bytes() {
    return NS == 1 ? MX_PART0 : (S1F == 4 ? MX_PART0 + 1024u : 2u * MX_PART0);
}

template <int CT, int RT, int NS, int S1F, int TH>
__global__ void __launch_bounds__(TH) mx_kernel(MxArgs a, uint32_t rows_per_block, uint32_t n_rowblocks, uint32_t grid_lg) {
    extern __shared__ uint4 mlds[];  // [n_steps][CT][step bytes], colc[3][CT*16] (alpha, -, column index), per-wave row-term exchange
    constexpr uint32_t SB = mx_step_bytes<NS, S1F>();
    constexpr int SLOTS = CT * 16;
    constexpr int NW = RT / 4;  // 64-row bitmap words per wave pass
    uint32_t rb = blockIdx.x, lg0 = 0, lg1 = a.n_lgroups;
    if (grid_lg) {  // every (row block, LDS group) pair is a block; the groups of a row block run next to each other on one XCD
        const uint32_t idx = blockIdx.x >> 3;
        rb = (idx / a.n_lgroups) * 8u + (blockIdx.x & 7u);
        lg0 = idx % a.n_lgroups;
        lg1 = lg0 + 1u;
    }
    const bool persist = KGWAS_MX_PERSIST && !grid_lg;
    if (!persist && rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t kb = lane >> 4, m = lane & 15u;
    const uint32_t n_steps = 4u * a.n_full + a.n_quarter;
    const uint32_t group_bytes = n_steps * CT * SB;
    char* lds = reinterpret_cast<char*>(mlds);
    float* colc = reinterpret_cast<float*>(lds + group_bytes);
    const int* colp = reinterpret_cast<const int*>(colc + 2 * SLOTS);
    float* wscr = colc + 3 * SLOTS + wave * (RT * 48u);  // wave-private: RT*16 x N1, RT*16 x (sqrt(d), E)
    const uint32_t rows_per_pass = (TH / 64) * (RT * 16u);
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    const int sc0 = (int)a.scale0;  // block scale of the first slice (E8M0 byte in every byte)
    uint32_t tested_local = 0;

    for (uint32_t lg = lg0; lg < lg1; lg++) {
        if (lg != lg0) __syncthreads();
        {
            const uint4* src = reinterpret_cast<const uint4*>(a.Bq + (size_t)lg * group_bytes);
            for (uint32_t i = threadIdx.x; i < group_bytes / 16u; i += TH) mlds[i] = src[i];
            if (threadIdx.x < SLOTS) {
                const CoarseCol cc = a.cols[lg * SLOTS + threadIdx.x];
                float al = __builtin_huge_valf();  // padding / N1 column: nothing survives
                if (cc.pheno >= 0) al = (float)(sqrt(a.thr[cc.pheno]) * cc.kalpha);  // NaN threshold (frozen column) -> NaN -> nothing survives
                colc[threadIdx.x] = al;
                colc[SLOTS + threadIdx.x] = cc.iu;
                reinterpret_cast<int*>(colc + 2 * SLOTS)[threadIdx.x] = cc.pheno;
            }
        }
        __syncthreads();

        uint32_t ro[RT];  // 32-bit byte offsets of this lane's rows (launch_mx guarantees the chunk spans < 4 GiB)
        const uint64_t wave_row0 = blk_row0 + wave * (RT * 16u);
        auto set_rows = [&](uint32_t (&o)[RT], uint64_t rb0) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint64_t r = rb0 + rt * 16u + m;
                if (r >= a.n_rows) r = a.n_rows - 1;
                o[rt] = ((uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u;
            }
        };
        // This lane's 16 bytes of full group g of each of its RT rows: ONE 16-byte load per row tile (the four kb-lanes of a
        // row fetch the group's 64 bytes as one contiguous piece; rows are 8-byte aligned). A full group's bytes always
        // exist in the row: the host counts only whole 512-sample groups as full (n_full = S / 512).
        struct __attribute__((aligned(8))) U4 {
            uint32_t x, y, z, w;
        };
        auto load_group = [&](uint32_t (&pc)[RT][4], const uint32_t (&o)[RT], uint32_t g) {
            const uint32_t b0 = 64u * g + 16u * kb;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint32_t off = o[rt] + b0;
                asm volatile("" : "+v"(off));  // a 32-bit offset on the scalar base, made here: not a hoisted (and spilled) 64-bit pointer
                U4 v;
                if (KGWAS_MX_ABLATE & 8)
                    v = U4{off, lane * 2654435761u, lane, b0};
                else {
#if KGWAS_MX_NT
                    typedef uint32_t nt_u4 __attribute__((ext_vector_type(4), aligned(8)));
                    const nt_u4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(rows_base + off));
                    v = U4{t.x, t.y, t.z, t.w};
#else
                    v = *reinterpret_cast<const U4*>(rows_base + off);
#endif
                }
                pc[rt][0] = v.x;
                pc[rt][1] = v.y;
                pc[rt][2] = v.z;
                pc[rt][3] = v.w;
            }
        };
        // Slice operands of the column tiles, read from LDS one UNIT (two column tiles, 16 MFMAs with two slices) ahead of
        // their MFMAs: while unit u's MFMAs run (256+ cycles), the ds_reads of unit u + 1 - the next two tiles of the step
        // or the first two of the next step - are in flight, into the registers unit u - 1 freed. No MFMA waits for a read
        // issued just before it, and only four tiles' operands (40 registers) are live at a time. The order is pinned with
        // scheduling barriers; inside a unit the compiler interleaves freely.
        constexpr int UT = 2;  // column tiles per unit (one: the same time, three or four: spills)
        constexpr int NU = (CT + UT - 1) / UT;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x4 Bx[CT], Bz[CT];  // dwords 0-3 of the first slice's operand, the second slice's
        u32x2 By[CT], Bv[CT];  // dwords 4-5 (FP6 operands only)
#pragma unroll
        for (int t = 0; t < CT; t++) Bx[t] = Bz[t] = (u32x4){0u, 0u, 0u, 0u}, By[t] = Bv[t] = (u32x2){0u, 0u};
        // A step's operands are read through three lane addresses (16-byte pieces; 8-byte pieces of the even and of the odd
        // tiles - two bases, so that the two 8-byte reads of a unit are not merged into one ds_read2 whose four result
        // registers then have to be moved next to their 16-byte halves), every read an immediate offset from them (a step's
        // operands span < 64 KB; left alone the compiler forms each read's address with a vector add of its own).
        struct StepAddr {
            uint32_t o16, o8, o8b;
        };
        auto step_addr = [&](const char* bstep) {
            StepAddr sa;
            sa.o16 = (uint32_t)(bstep - lds) + lane * 16u;
            sa.o8 = (uint32_t)(bstep - lds) + lane * 8u + 1024u;
            sa.o8b = sa.o8 + SB;
            asm volatile("" : "+v"(sa.o16), "+v"(sa.o8), "+v"(sa.o8b));
            return sa;
        };
        auto read_unit = [&](int u, const StepAddr& sa) {
#pragma unroll
            for (int t = UT * u; t < (UT * u + UT < CT ? UT * u + UT : CT); t++) {
                if (KGWAS_MX_ABLATE & 4) {
                    Bx[t] = (u32x4){lane, (uint32_t)t, 3u, (uint32_t)u};
                    By[t] = (u32x2){5u, 6u};
                    Bz[t] = (u32x4){lane, (uint32_t)t, 7u, (uint32_t)u};
                    Bv[t] = (u32x2){1u, 2u};
                    continue;
                }
                const char* p16 = lds + sa.o16 + t * SB;
                const char* p8 = (t & 1) ? lds + sa.o8b + (t - 1) * SB : lds + sa.o8 + t * SB;
                Bx[t] = *reinterpret_cast<const u32x4*>(p16);
                By[t] = *reinterpret_cast<const u32x2*>(p8);
                if (NS == 2) {
                    Bz[t] = *reinterpret_cast<const u32x4*>(p16 + MX_PART0);
                    if (S1F == 2) Bv[t] = *reinterpret_cast<const u32x2*>(p8 + MX_PART0);
                }
            }
        };
        uint32_t piece[RT][4];
        // the rows of this wave's pass ps: consecutive passes of a block, or - persistent blocks - every gridDim.x-th pass of the chunk
        const uint64_t pass_stride = persist ? (uint64_t)gridDim.x * rows_per_pass : rows_per_pass;
        const uint64_t wave_row00 = persist ? (uint64_t)blockIdx.x * rows_per_pass + wave * (RT * 16u) : wave_row0;
        const uint64_t n_pass_blk = persist ? ~0ull : (rows_per_block + rows_per_pass - 1) / rows_per_pass;
        set_rows(ro, wave_row00);
        if (a.n_full && wave_row00 < a.n_rows) load_group(piece, ro, 0);
        StepAddr sadr = step_addr(lds);
        read_unit(0, sadr);  // step 0 of the first pass; every pass's last step fetches it for the next
        for (uint64_t ps = 0; ps < n_pass_blk; ps++) {
            const uint64_t rbase = wave_row00 + ps * pass_stride;
            if (rbase >= a.n_rows) break;  // wave-uniform
            uint32_t ro_next[RT];
            set_rows(ro_next, rbase + pass_stride);
#if KGWAS_MX_WARM
            {
                // rows [rbase + rows_per_pass, + RT*16) of this wave: a contiguous span of the table (whatever the stride)
                const uint64_t r0 = rbase + pass_stride;
                if (r0 < a.n_rows) {  // wave-uniform
                    const uint64_t r1 = r0 + RT * 16u < a.n_rows ? r0 + RT * 16u : a.n_rows;
                    const uint32_t b0 = ((uint32_t)r0 * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u & ~127u;
                    const uint32_t b1 = ((uint32_t)(r1 - 1u) * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u + avail_b - 4u;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this pass's first pieces: requested before the last epilogue)
                    const uint32_t junk = (uint32_t)(reinterpret_cast<char*>(colc + 3 * SLOTS + (TH / 64) * (RT * 48u)) - lds);
                    const uint32_t n_warm = (b1 - b0) / 8192u + 1u;
                    for (uint32_t k = 0; k < n_warm; k++) {
                        uint32_t off = (b1 - b0) - k * 8192u >= lane * 128u ? b0 + k * 8192u + lane * 128u : b1;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(junk), "v"(off), "s"(rows_base) : "memory");
                    }
                }
            }
#endif
            mxv4f acc[RT][CT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int t = 0; t < CT; t++) acc[rt][t] = (mxv4f){0.0f, 0.0f, 0.0f, 0.0f};

            // One unit: RT MFMAs per slice for each of its column tiles - both slices' products into the same accumulator
            // (block scale sc0 on the first slice, 2^0 on the second).
            auto mfma_unit = [&](int u, const mxv8i (&A)[RT], int sa) {
#pragma unroll
                for (int t = UT * u; t < (UT * u + UT < CT ? UT * u + UT : CT); t++) {
                    const mxv8i B0 = {(int)Bx[t].x, (int)Bx[t].y, (int)Bx[t].z, (int)Bx[t].w, (int)By[t].x, (int)By[t].y, 0, 0};
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
                        acc[rt][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[rt], B0, acc[rt][t], 4, 2, 0, sa, 0, sc0);
                    if (NS == 2) {
                        mxv8i B1 = {(int)Bz[t].x, (int)Bz[t].y, (int)Bz[t].z, (int)Bz[t].w, 0, 0, 0, 0};
                        if (S1F == 2) {
                            B1[4] = (int)Bv[t].x;
                            B1[5] = (int)Bv[t].y;
                        }
#pragma unroll
                        for (int rt = 0; rt < RT; rt++)
                            acc[rt][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[rt], B1, acc[rt][t], 4, S1F, 0, sa, 0, 0x7F7F7F7F);
                    }
                }
            };
            // step = [reads of unit 1 | MFMAs of unit 0 | reads of unit 2 | MFMAs of unit 1 | ... | reads of the next step's unit 0 | MFMAs of the last unit]
            auto run_step = [&](const mxv8i (&A)[RT], const char* bs_next, int sa) {
                const StepAddr nadr = step_addr(bs_next);
                if (NU == 1) {
                    // one unit per step: the next step's operands cannot land in the registers this step still multiplies
                    // with - they go to a second set (at most two tiles) and move over afterwards
                    u32x4 kx[CT], kz[CT];
                    u32x2 ky[CT], kv[CT];
#pragma unroll
                    for (int t = 0; t < CT; t++) kx[t] = Bx[t], ky[t] = By[t], kz[t] = Bz[t], kv[t] = Bv[t];
                    read_unit(0, nadr);
                    u32x4 nx[CT], nz[CT];
                    u32x2 ny[CT], nv[CT];
#pragma unroll
                    for (int t = 0; t < CT; t++) nx[t] = Bx[t], ny[t] = By[t], nz[t] = Bz[t], nv[t] = Bv[t];
#pragma unroll
                    for (int t = 0; t < CT; t++) Bx[t] = kx[t], By[t] = ky[t], Bz[t] = kz[t], Bv[t] = kv[t];
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_unit(0, A, sa);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < CT; t++) Bx[t] = nx[t], By[t] = ny[t], Bz[t] = nz[t], Bv[t] = nv[t];
                } else {
#pragma unroll
                    for (int u = 0; u < NU; u++) {
                        if (u + 1 < NU)
                            read_unit(u + 1, sadr);
                        else
                            read_unit(0, nadr);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_unit(u, A, sa);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                sadr = nadr;
            };

            // full 512-sample groups: four steps over this lane's 16 bytes of each of its RT rows. The pieces of the NEXT
            // group - the next one of this pass or, from the last one, group 0 of the wave's next rows - are requested as
            // soon as step 3 has expanded the current ones: a whole step (and, across passes, the epilogue) ahead.
#if KGWAS_MX_PRIO || KGWAS_MX_PRIO_EPI
            __builtin_amdgcn_s_setprio(KGWAS_MX_PRIO);
#endif
            for (uint32_t g = 0; g < a.n_full; g++) {
                const char* bg = lds + (size_t)g * 4u * CT * SB;
                const bool last_g = g + 1u == a.n_full;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    mxv8i A[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        A[rt] = (mxv8i){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            if (KGWAS_MX_ABLATE & 2)
                                A[rt][q] = (int)piece[rt][q];
                            else
                                A[rt][q] = j < 3 ? (int)(piece[rt][q] & (0x11111111u << j)) : (int)((piece[rt][q] >> 1) & 0x44444444u);
                        }
                    }
                    const char* bs = bg + j * CT * SB;
                    const char* bs_next = bs + CT * SB;
                    if (j == 3) {
                        // (selects, not branches: a branch here splits the loop body, the MFMAs of steps 0-2 sink below it
                        // and their operands - three steps' worth - are all live across it)
                        __builtin_amdgcn_sched_barrier(0);
                        uint32_t on[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) on[rt] = last_g ? ro_next[rt] : ro[rt];
                        load_group(piece, on, last_g ? 0u : g + 1u);
                        if (last_g && a.n_quarter == 0) bs_next = lds;  // the pass's last step: step 0 of the next pass
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // nibble bit 0 / 1 / 2 / 2 (bit 3 shifted down): 0.5 / 1.0 / 2.0 / 2.0 x 2^0 / 2^-1 / 2^-2 / 2^-2 = 0.5
                    run_step(A, bs_next, j == 0 ? 0x7F7F7F7F : j == 1 ? 0x7E7E7E7E : 0x7D7D7D7D);
                }
            }
            // quarter groups: 128 samples per step, the lane's own dword shifted by 0..3
            for (uint32_t x = 0; x < a.n_quarter; x++) {
                uint32_t b0 = 64u * a.n_full + 16u * x + 4u * kb;
                b0 = b0 + 4u <= avail_b ? b0 : avail_b - 4u;
                mxv8i A[RT];
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const uint32_t w = (KGWAS_MX_ABLATE & 8) ? ro[rt] + b0 : *reinterpret_cast<const uint32_t*>(rows_base + (ro[rt] + b0));
                    A[rt] = (mxv8i){(int)(w & 0x11111111u), (int)((w >> 1) & 0x11111111u), (int)((w >> 2) & 0x11111111u),
                                    (int)((w >> 3) & 0x11111111u), 0, 0, 0, 0};
                }
                const char* bs = lds + (size_t)(4u * a.n_full + x) * CT * SB;
                run_step(A, x + 1u == a.n_quarter ? lds : bs + CT * SB, 0x7F7F7F7F);
            }

#if KGWAS_MX_PRIO || KGWAS_MX_PRIO_EPI
            // The pass's epilogue at raised priority, the main loop at the default one: a wave in its epilogue issues vector
            // instructions only, and the sooner it is through them the sooner the SIMD has two MFMA streams again
            // (-1.8 % in alternating runs, tools/mx_ab.sh; raising the MAIN loop's priority instead: +-0).
            __builtin_amdgcn_s_setprio(KGWAS_MX_PRIO_EPI);
#endif
            // Per-row terms (as in score_coarse.hip). Lane (kb, m) holds accumulator registers of the rows kb*4 + jj of
            // its RT row tiles ("row slot" i = rt*4 + jj); the 16 m-lanes of a kb share them. N1 comes from the ones column
            // (slot 15 of the last column tile: lane (kb, 15) holds it for all its slots); through a wave-private LDS
            // exchange every lane computes the terms of ONE row per 64 rows - sqrt(d) rounded down, +inf for a row that does
            // not exist or fails the MAC filter (nothing survives), and the row's error term E(N1) in accumulator units,
            // rounded up - and reads back the ones it needs.
            float2* trm = reinterpret_cast<float2*>(wscr + RT * 16);
            {
                float* n1s = wscr;
                if (m == 15u) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) *reinterpret_cast<mxv4f*>(n1s + kb * (RT * 4u) + rt * 4) = acc[rt][CT - 1];
                }
                __builtin_amdgcn_wave_barrier();
                const uint64_t left = a.n_rows - rbase;
                const uint32_t rows_here = left < RT * 16u ? (uint32_t)left : RT * 16u;
                const bool mac_any = a.S >= 2u * a.min_count;  // else no N1 can satisfy mc <= N1 <= S - mc
                const uint32_t span = a.S - 2u * a.min_count;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    // entry e = lane + 64 w of the exchange area = (kb' = e / (RT*4), slot i = e % (RT*4)): row 16 (i / 4) + 4 kb' + i % 4
                    const uint32_t e = lane + 64u * w;
                    const uint32_t kbe = e / (RT * 4u), ie = e % (RT * 4u);
                    const uint32_t row = 16u * (ie >> 2) + 4u * kbe + (ie & 3u);
                    const float f = n1s[e];
                    const uint32_t n1r = (uint32_t)f;
                    const bool ok = mac_any & (row < rows_here) & ((n1r - a.min_count) <= span);
                    if (lg == 0) tested_local += ok ? 1u : 0u;
                    const float sq = __builtin_amdgcn_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; 1 ulp sqrt; (1 - 2^-20)
                    float2 tm;
                    tm.x = ok ? sq : __builtin_huge_valf();
                    tm.y = (a.eg_max + fminf(a.rall_max, f * a.rmax_max)) * 1.000001f;
                    trm[e] = tm;
                }
                __builtin_amdgcn_wave_barrier();
            }
            // The test. A pair (row slot i, column p) survives iff  |acc| - alpha_p * sqrt(d_i) + E_i >= 0  (one fma with |.|
            // as an operand modifier, one add; the roundings are covered by the constants' safety margins, see above). It is
            // only evaluated for row slots that can have a survivor at all: with alpha_min the smallest alpha of the
            // lane's columns, max_p |acc_p| - alpha_min * sqrt(d_i) + E_i >= 0 is necessary (the fma is monotone in alpha) -
            // a running maximum of |acc| per row slot (v_max3_f32 with |.| modifiers: half a lane-op per pair) and ONE fma
            // + compare per row slot instead of an fma per pair. Permutation columns share their value set, so their
            // alphas differ by the few per cent their thresholds do, and the pre-test flags ~1.2x the row slots the exact
            // one would; whatever the columns, only the amount of rare-path work depends on how far the alphas spread.

            float alc[CT];
            float al_min = __builtin_huge_valf();
#pragma unroll
            for (int t = 0; t < CT; t++) {
                alc[t] = colc[t * 16 + m];
                al_min = fminf(al_min, alc[t]);  // (a NaN alpha - frozen column - is skipped; +inf = padding / ones column)
            }
            const bool ones_lane = m == 15u;  // slot 15 of the last tile is the ones column: its accumulator (N1) is no margin
#pragma unroll
            for (int w = 0; w < NW; w++) {
                float sqd[16], er[16];  // the terms of this word's 16 row slots
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float4 v = *reinterpret_cast<const float4*>(trm + kb * (RT * 4u) + 16 * w + i);
                    sqd[i] = v.x;
                    er[i] = v.y;
                    sqd[i + 1] = v.z;
                    er[i + 1] = v.w;
                }
                auto pair_margin = [&](int i, int t, float al) {
                    const int gi = 16 * w + i;
                    return fmaf(-al, sqd[i], fabsf(acc[gi >> 2][t][gi & 3]));  // NaN (frozen column) never passes
                };
                uint64_t hit[16];
                if (!(KGWAS_MX_ABLATE & 1)) {
                    float mx[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int gi = 16 * w + i;
                        const float v = acc[gi >> 2][CT - 1][gi & 3];
                        mx[i] = ones_lane ? 0.0f : fabsf(v);
                    }
#pragma unroll
                    for (int t = 0; t + 1 < CT; t++)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int gi = 16 * w + i;
                            mx[i] = fmaxf(mx[i], fabsf(acc[gi >> 2][t][gi & 3]));
                        }
#pragma unroll
                    for (int i = 0; i < 16; i++) hit[i] = __ballot(fmaf(-al_min, sqd[i], mx[i]) + er[i] >= 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) hit[i] = 0;
                    int x = 0;  // keeps every accumulator (and its MFMAs) alive at one lane-op each
#pragma unroll
                    for (int i = 0; i < 16; i++)
#pragma unroll
                        for (int t = 0; t < CT; t++) x ^= __float_as_int(acc[(16 * w + i) >> 2][t][i & 3]);
                    hit[0] = __ballot(x == 0x7fffffff);
                }
                uint64_t hit_any = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) hit_any |= hit[i];
                if ((KGWAS_MX_ABLATE & 32) && hit_any) {
                    if (lane == 0) atomicAdd(&a.tested[0], 0ull);
                    hit_any = 0;
                }
                if (hit_any) {  // wave-uniform
                    // mb[t] bit i = pair (row slot 16 w + i, column t*16 + m) survives; worked out for the row slots that had a hit.
                    uint32_t mb[CT];
#pragma unroll
                    for (int t = 0; t < CT; t++) mb[t] = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        if (hit[i]) {  // wave-uniform
#pragma unroll
                            for (int t = 0; t < CT; t++) {
                                float al = alc[t];
                                asm volatile("" : "+v"(al));
                                mb[t] |= (pair_margin(i, t, al) + er[i] >= 0.0f) ? (1u << i) : 0u;
                            }
                        }
                    }
                    // Column (t, m)'s 64 row bits of this word sit in four lanes (kb = 0..3; bit i = 4 rt' + jj of mb[t] is row
                    // rt' * 16 + 4 kb + jj of the word's 64). Each lane stores its 16 bits as quarter kb of the column's
                    // word: no cross-lane traffic; launch_bitmap_keys(nibble_transposed = true) puts the nibbles back in
                    // row order.
                    unsigned short* bm16 = reinterpret_cast<unsigned short*>(a.bitmap) + ((rbase >> 6) + w) * 4u + kb;
#pragma unroll
                    for (int t = 0; t < CT; t++)
                        if (mb[t]) bm16[(uint64_t)colp[t * 16 + m] * a.words_per_col * 4u] = (unsigned short)mb[t];  // column >= 0 wherever a bit is set
                }
            }
            __builtin_amdgcn_wave_barrier();  // the exchange area is rewritten by the next pass
#pragma unroll
            for (int rt = 0; rt < RT; rt++) ro[rt] = ro_next[rt];
        }
    }
#if KGWAS_MX_WARM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no transfer may outlive the block's LDS
#endif
    if (a.tested) {
        uint32_t v = tested_local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// slice operands of one LDS group + the group's per-column constants (3 x up to 112 words) + the waves' row-term
// exchange areas (RT*48 words each: 1536 B per wave at RT = 8)
size_t mx_lds_bytes(uint32_t n_steps, uint32_t CT, uint32_t n_slices, uint32_t s1_fp6) {
    const uint32_t sb = n_slices == 1 ? MX_PART0 : (s1_fp6 ? 2u * MX_PART0 : MX_PART0 + 1024u);
    return (size_t)n_steps * CT * sb + 3u * 112u * 4u + 12u * 768u + 256u;  // (twelve waves where CT <= 3; 256: junk words of KGWAS_MX_WARM)
}
uint32_t mx_step_bytes_rt(uint32_t n_slices, uint32_t s1_fp6) { return n_slices == 1 ? MX_PART0 : (s1_fp6 ? 2u * MX_PART0 : MX_PART0 + 1024u); }
uint32_t mx_row_tiles(uint32_t CT) { return 4u; }

template <int CT, int RT, int NS, int S1F, int TH = 512>
static hipError_t launch_mx_t(const MxArgs& a, uint32_t rows_per_block, size_t lds, hipStream_t st) {
    const uint32_t rpp = (TH / 64) * RT * 16u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mx_kernel<CT, RT, NS, S1F, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t grid_lg = a.n_lgroups > 1 ? 1u : 0u;
    uint32_t grid = grid_lg ? (n_rowblocks + 7u) / 8u * 8u * a.n_lgroups : n_rowblocks;
    if (KGWAS_MX_PERSIST && !grid_lg) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t pr;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        }
        const uint64_t n_pass = (a.n_rows + rpp - 1) / rpp;
        grid = (uint32_t)std::min<uint64_t>(n_pass, (uint64_t)n_cu);
    }
    launch_last(mx_kernel<CT, RT, NS, S1F, TH>, dim3(grid), dim3(TH), lds, st, a, rows_per_block, n_rowblocks, grid_lg);
    return hipGetLastError();
}

template <int NS, int S1F>
static hipError_t launch_mx_ct(const MxArgs& a, uint32_t CT, uint32_t rows_per_block, size_t lds, hipStream_t st) {
    switch (CT) {
        // up to three column tiles (48 accumulator registers with four row tiles): 168 registers, three waves per SIMD
        case 1: return launch_mx_t<1, 4, NS, S1F, 768>(a, rows_per_block, lds, st);
        case 2: return launch_mx_t<2, 4, NS, S1F, 768>(a, rows_per_block, lds, st);
        case 3: return launch_mx_t<3, 4, NS, S1F, 768>(a, rows_per_block, lds, st);
        case 4: return launch_mx_t<4, 4, NS, S1F>(a, rows_per_block, lds, st);
        case 5: return launch_mx_t<5, 4, NS, S1F>(a, rows_per_block, lds, st);
        case 6: return launch_mx_t<6, 4, NS, S1F>(a, rows_per_block, lds, st);
        case 7: return launch_mx_t<7, 4, NS, S1F>(a, rows_per_block, lds, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mx(const MxArgs& a, uint32_t CT, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const size_t lds = mx_lds_bytes(4u * a.n_full + a.n_quarter, CT, a.n_slices, a.s1_fp6);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if ((a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit byte offsets
#ifdef KGWAS_MX_BENCH_ONLY  // experiments: only the shape of the 1024 x 101 bench (fast compiles)
    if (a.n_slices == 2 && !a.s1_fp6 && CT == 7) return launch_mx_t<7, 4, 2, 4>(a, rows_per_block, lds, st);
    return hipErrorInvalidValue;
#else
    if (a.n_slices == 1) return launch_mx_ct<1, 4>(a, CT, rows_per_block, lds, st);
    if (a.n_slices == 2 && !a.s1_fp6) return launch_mx_ct<2, 4>(a, CT, rows_per_block, lds, st);
    if (a.n_slices == 2 && a.s1_fp6) return launch_mx_ct<2, 2>(a, CT, rows_per_block, lds, st);
    return hipErrorInvalidValue;
#endif
}

}  // namespace kgwas
