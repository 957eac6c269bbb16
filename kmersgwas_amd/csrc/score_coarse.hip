// score_coarse.hip — coarse int8-MFMA filter (XDL matrix pipe) + exact re-scoring of the survivors.
//
// The exact scorers (score_mfma.hip / score_valu.hip) reproduce calculate_kmer_score
// (src/kmers_multiple_databases.cpp:327-363) bit for bit, but the exact float32 order is capped by the
// vector-rate f32 MFMA (157 TFLOP/s). Once the heaps are full only ~1e-4 of the (k-mer, column) pairs
// can beat a column's threshold, so the sparse phase is split in two:
//
//  1. coarse_kernel — for EVERY pair, an integer approximation of yigi = sum_i g_i*y_i on
//     v_mfma_i32_16x16x64_i8 (XDL pipe, ~25x the f32 rate, runs beside the VALU):
//        y_i ~ s0*q0_i + s1*q1_i,  q0, q1 in [-127, 127]  (two int8 slices per phenotype column)
//        yc  = s0*D0 + s1*D1,      D0, D1 = exact int32 dot products of the bit row with q0, q1
//     and a RIGOROUS bound E_p >= |yigi_ref - yc|, where yigi_ref is the value the reference's float32
//     chain produces:   |yigi_ref - sum g_i y_i| <= gamma_{L/4+3} * sum|y_i|   (float32 summation)
//                       |sum g_i y_i  - yc|      <= sum_i |y_i - s0 q0_i - s1 q1_i|   (quantisation)
//     With r = N*yigi - N1*sum (same `sum` as the reference), |r_ref| <= |r_c| + N*E_p, so a pair can
//     only satisfy score_ref > thr if (|r_c| + N*E_p)^2 >= thr*d. Everything else is provably below the
//     threshold and is dropped; the rest ("survivors", true candidates plus a ~1e-6 fringe) is listed.
//  2. rescore_kernel — the survivors are re-scored in the exact reference order on the VALU (one lane per
//     survivor, y wave-uniform) and go through the same finish_pair as the exact scorers: exact test
//     against thr, threshold histogram, candidate record. Results are therefore bit-identical to the
//     exact path; the coarse pass only decides what is worth looking at.
//
// Coarse mode does not care about accumulation order, so the k-index of the MFMA is mapped to samples in
// the way that needs no data exchange and the fewest lane-ops: in a 512-sample group g, lane (m = lane&15,
// kg = lane>>4) loads ITS OWN 16 bytes (dwords q = 0..3) of row m (samples 512g+128kg .. +127), and MFMA step
// j = 0..7 takes bit 8e + j of dword q as k-element 4q + e of that lane, i.e.
//        k = 16*kg + 4q + e   <->   sample 512g + 128kg + 32q + 8e + j.
// The four kg-lanes of a row thus fetch 64 contiguous bytes per instruction, every byte is used by the lane that
// loaded it, and an operand dword is (dword >> j) & 0x01010101: two lane-ops per 4 operand bytes (the earlier
// nibble * 0x00204081 & 0x01010101 needed three, and the vector instructions beside the MFMAs are what bounds this
// kernel); each expanded operand feeds T column tiles, each B operand (ds_read_b128) four row tiles.
#include <algorithm>
#include <type_traits>

#include "score_common.h"

#ifndef KGWAS_COARSE_PHASES
#define KGWAS_COARSE_PHASES 1  // pin the per-step order: LDS reads, operand expansion, MFMAs
#endif
// Timing experiments only (wrong results; tools/coarse_variants.sh builds the variants). Bits: 1 the tests are replaced by
// an XOR over all accumulators (every MFMA stays alive), 2 no operand expansion, 4 no LDS operand reads, 8 no row loads,
// 16 constant row terms, 32 a token side effect instead of the survivor emission.
#ifndef KGWAS_COARSE_ABLATE
#define KGWAS_COARSE_ABLATE 0
#endif
#ifndef KGWAS_RESCORE_HALVES
#define KGWAS_RESCORE_HALVES 0  // experiments (rescore_block): with KGWAS_RESCORE_WAVES, tools/rescore_variants.sh - none ahead, see there
#endif
#ifndef KGWAS_RESCORE_FMA
#define KGWAS_RESCORE_FMA 1  // 0: the select-and-add form of rounds 1-2 (2.5 lane-ops per sample)
#endif

namespace kgwas {

typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef KGWAS_COARSE_TIMELINE  // diagnostics build (tools/coarse_timeline.py): cycle stamps of waves 0 and 4 of one block
__device__ unsigned long long g_coarse_tl[2 * 16 * 64];
#define TL_STAMP(idx)                                                                       \
    do {                                                                                    \
        if (tl_on) {                                                                        \
            const unsigned long long t_ = clock64();                                        \
            if (lane == 0) g_coarse_tl[tl_base + (idx)] = t_;                               \
        }                                                                                   \
    } while (0)
#else
#define TL_STAMP(idx) do { } while (0)
#endif

// A operand of step j from the lane's four row dwords: byte e of operand dword q = bit 8e + j of row dword q,
// i.e. k-element 4q + e <-> sample 32q + 8e + j of the lane's 128 (two lane-ops per dword, one for j = 0).
__device__ __forceinline__ i32x4 expand_step(const uint32_t (&w)[4], int j) {
    i32x4 r;
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = (int)((w[q] >> j) & 0x01010101u);
    return r;
}

// T  = operand tiles (16 columns of one int8 slice each) per LDS group,
// NS = int8 slices per column (1: tile t of a group = column tile; 2: tiles 2g, 2g+1 = the two slices of column group g).
//
// The conservative test. The host centres the quantisation at c = sum/N (sum = the reference's float32 sum), so
//   r_c = N*yc - N1*sum = N*u*Dc            Dc = D0 (one slice) or 254*D0 + D1 (two slices, s0 = 254*s1), u = s0 or s1,
// an exact integer times a column constant. A pair can only have score_ref > thr if
//   |Dc| >= alpha_p * sqrt(d(N1)) - E_p(N1)/u_p,    alpha_p = sqrt(thr_p)/(N*u_p),  E_p(N1) = Eg_p + min(Rall_p, N1*rmax_p).
// The error terms are taken in units of Dc (Eg_p/u_p, ... : the quantisation residual is at most half a unit per
// sample whatever the column's scale) and as their maxima over all columns, so E/u is a per-ROW term and the test is
//   margin = |Dc| - alpha_p * sqrt(d)   (float(Dc), then one float32 fma),    survivor iff margin + E >= 0
// evaluated from constants the host rounded in the safe direction (alpha down by 2^-19 relative, sqrt(d) down by
// 2^-20, the error terms up by 1e-6): the slack 2.8e-6 * alpha * sqrt(d) dominates the roundings of float(Dc) and of
// the fma (6e-8 relative each), and the final addition cannot change the sign of an exact non-negative sum - so no
// possible candidate is dropped. The maximum of the margins of a lane's pairs of one row is tested once.
//
// N1 comes out of the matrix pipe as well: the last operand column of every LDS group holds 1 for each phenotyped
// sample (0 elsewhere), so its dot product IS the masked popcount - no masks, no popcounts, no mask loads in the
// main loop, and bits of unphenotyped samples or row padding meet zeros in every operand column.
template <int T, int NS, int TH = 512>
__global__ void __launch_bounds__(TH) coarse_kernel(CoarseArgs a, uint32_t rows_per_block, uint32_t n_rowblocks, uint32_t grid_lg) {
    extern __shared__ i32x4 blds[];  // [n_kgroups][8][T][64] x 16 bytes, colc[3][PG*16] (alpha, 1/u, column index), survivor buffer
    constexpr int PG = T / NS;       // column groups (of 16) per LDS group
    constexpr int RT = 4;
    constexpr int SLOTS = PG * 16;
    // Several LDS groups: either this block walks them one after the other over its rows (grid_lg = 0: every pass over
    // the rows re-reads them, 1 MB per block and pass, far more than the L2 keeps across a pass) or every (row block,
    // group) pair is a block of its own (grid_lg = 1). Blocks go to XCD (id % 8) in id order, so the groups of a row
    // block get consecutive ids on ONE XCD: they start within a few per cent of a block's run time of each other and
    // stream through the same rows together - one of them fetches a row from HBM, the others find it in the L2.
    uint32_t rb = blockIdx.x, lg0 = 0, lg1 = a.n_lgroups;
    if (grid_lg) {
        const uint32_t idx = blockIdx.x >> 3;
        rb = (idx / a.n_lgroups) * 8u + (blockIdx.x & 7u);
        lg0 = idx % a.n_lgroups;
        lg1 = lg0 + 1u;
    }
    if (rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t kg = lane >> 4, m = lane & 15u;
    const uint32_t n_kgroups = a.n_kgroups;
    const uint32_t group_vec = n_kgroups * 8u * T * 64u;  // i32x4 elements per LDS group
    float* colc = reinterpret_cast<float*>(blds + group_vec);
    const int* colp = reinterpret_cast<const int*>(colc + 2 * SLOTS);
    // Survivors leave as a bitmap: one 64-bit word per (phenotype column, 64 rows of a wave pass), bit = row of the pass
    // (the chunk's bitmap is zeroed beforehand; only non-zero words are stored). Row order is then a property of the
    // bitmap and the ordered per-column key lists come out of a popcount scan (launch_bitmap_keys) - the first version
    // appended keys to one global list (per-block LDS buffering, one device atomic per block) that a radix sort then
    // had to put in (column, row) order: 4.6 ms of a 22 ms pass at 101 columns.
    float* wscr = colc + 3 * SLOTS + wave * 192u;  // wave-private: 64 x N1, 64 x (sqrt(d), E)
    const uint32_t rows_per_pass = (blockDim.x >> 6) * (RT * 16u);
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    uint32_t tested_local = 0;

    for (uint32_t lg = lg0; lg < lg1; lg++) {
        if (lg != lg0) __syncthreads();
        {
            const i32x4* src = reinterpret_cast<const i32x4*>(a.Bq) + (size_t)lg * group_vec;
            for (uint32_t i = threadIdx.x; i < group_vec; i += blockDim.x) blds[i] = src[i];
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

        // piece[rt][q] = dword q of this lane's 16 bytes of row tile rt for the current (pass, sample group) unit
        // (bytes 64g + 16kg .. +15 of the row's bits; beyond the row's data the load is clamped onto its last 8 bytes:
        // whatever bits arrive there meet zero operands, sample slots >= S are zero in every column). The pieces of
        // the NEXT unit - the next sample group, or group 0 of the wave's next 64 rows - are fetched into the same
        // registers as soon as the current unit has expanded them: 4-5 steps (2000+ cycles) ahead of their use, and
        // across the epilogue.
        uint32_t piece[RT][4];
        uint32_t ro[RT];  // 32-bit byte offsets (launch_coarse guarantees the chunk spans < 4 GiB) of the rows to fetch next
        auto set_rows = [&](uint64_t rb0) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint64_t r = rb0 + rt * 16u + m;
                if (r >= a.n_rows) r = a.n_rows - 1;
                ro[rt] = ((uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u;
            }
        };
        auto load_half = [&](uint32_t g, int h) {
            uint32_t b0 = 64u * g + 8u * h + 16u * kg;
            b0 = b0 + 8u <= avail_b ? b0 : avail_b - 8u;  // unconditional load (a branch here splits the scheduling region)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint2 v;
                if (KGWAS_COARSE_ABLATE & 8)
                    v = make_uint2(ro[rt] + b0, lane * 2654435761u);
                else
                    v = *reinterpret_cast<const uint2*>(rows_base + (ro[rt] + b0));
                piece[rt][2 * h] = v.x;
                piece[rt][2 * h + 1] = v.y;
            }
        };
        const uint64_t wave_row0 = blk_row0 + wave * (RT * 16u);

        for (uint32_t ps = 0; ps * rows_per_pass < rows_per_block; ps++) {
            const uint64_t rbase = wave_row0 + (uint64_t)ps * rows_per_pass;
            if (rbase >= a.n_rows) break;  // wave-uniform
            i32x4 acc[RT][T];
#ifdef KGWAS_COARSE_TIMELINE
            const bool tl_on = blockIdx.x == KGWAS_COARSE_TIMELINE && (wave & 3u) == 0u && ps < 16u && lg == 0;
            const uint32_t tl_base = ((wave >> 2) * 16u + ps) * 64u;
#endif
            TL_STAMP(0);

            // One step = one bit position of every byte of the lane's 16 row bytes = 64 MFMA k-elements, T x RT MFMAs.
            // Per step the B operands are requested from LDS first, the A operands are expanded while those reads are in
            // flight, then the MFMAs issue back to back. What was measured on gfx950 about this loop (tools/issue_model.hip,
            // tools/coarse_variants.sh): beside each 16-cycle MFMA a SIMD issues two other vector instructions for free
            // and pays ~4 cycles for every further one; a long run of vector instructions in one wave (the epilogue) is
            // NOT hidden behind its partner's MFMAs, with or without s_setprio or a start offset between the two waves;
            // software-pipelined variants (A operands double-buffered and interleaved with the MFMAs, B operands reloaded
            // in place behind their tile, next row dwords prefetched) all measured 5-25 % slower than this plain order.
            auto run_group = [&](uint32_t g, auto first_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;  // first sample group: the accumulators start at zero
                const i32x4* bg = blds + (size_t)g * 8u * T * 64u + lane;
                if (FIRST) set_rows(rbase);
                load_half(g, 0);
                load_half(g, 1);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (g < 2) TL_STAMP(1 + (g * 8 + j) * 3);  // step start
                    i32x4 B[T];
#pragma unroll
                    for (int t = 0; t < T; t++) B[t] = (KGWAS_COARSE_ABLATE & 4) ? (i32x4){(int)(lane & 1u), 0, 1, 0} : bg[(j * T + t) * 64];
#if KGWAS_COARSE_PHASES
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    i32x4 A[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        if (KGWAS_COARSE_ABLATE & 2)
                            A[rt] = (i32x4){(int)(piece[rt][0] & 0x01010101u), (int)(piece[rt][1] & 0x01010101u),
                                            (int)(piece[rt][2] & 0x01010101u), (int)(piece[rt][3] & 0x01010101u)};
                        else
                            A[rt] = expand_step(piece[rt], j);
                    }
#if KGWAS_COARSE_PHASES
                    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef KGWAS_COARSE_TIMELINE
                    __builtin_amdgcn_s_waitcnt(0);  // operands arrived (vmcnt 0, lgkmcnt 0)
                    if (g < 2) TL_STAMP(2 + (g * 8 + j) * 3);  // operands ready, MFMAs start
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int t = 0; t < T; t++) {
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) {
                            if (FIRST && j == 0)
                                acc[rt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt], B[t], (i32x4){0, 0, 0, 0}, 0, 0, 0);
                            else
                                acc[rt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt], B[t], acc[rt][t], 0, 0, 0);
                        }
                    }
#if KGWAS_COARSE_PHASES
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    if (g < 2) TL_STAMP(3 + (g * 8 + j) * 3);  // MFMAs issued
                }
            };
            run_group(0u, std::true_type{});
            for (uint32_t g = 1; g < n_kgroups; g++) run_group(g, std::false_type{});

            TL_STAMP(50);  // main loop done
            // Per-row terms. Lane (kg, m) holds accumulator registers of the 16 rows kg*4 + jj of the four row tiles
            // ("row slot" i = rt*4 + jj); the 16 m-lanes of a kg share them. N1 comes from the ones column (slot 15
            // of the last column group: lane (kg, 15) holds it for all 16 slots); through a wave-private LDS
            // exchange every lane computes the terms of ONE row (slot i = m) - sqrt(d) rounded down, +inf for a row
            // that does not exist or fails the MAC filter (nothing survives), and the row's error term E(N1) in
            // units of Dc, rounded up - and reads back the 16 it needs (the same terms computed 16 times per lane
            // were a quarter of the kernel's vector work).
            float sqd[RT * 4], er[RT * 4];
            {
                int* n1s = reinterpret_cast<int*>(wscr);
                float2* trm = reinterpret_cast<float2*>(wscr + 64);
                if (m == 15u) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) *reinterpret_cast<i32x4*>(n1s + kg * 16u + rt * 4) = acc[rt][T - 1];
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t n1r = (uint32_t)n1s[lane];  // row slot m of this kg
                const uint64_t left = a.n_rows - rbase;
                const uint32_t rows_here = left < 64u ? (uint32_t)left : 64u;
                const bool mac_any = a.S >= 2u * a.min_count;  // else no N1 can satisfy mc <= N1 <= S - mc
                const uint32_t span = a.S - 2u * a.min_count;
                const bool ok = mac_any & ((m >> 2) * 16u + kg * 4u + (m & 3u) < rows_here) & ((n1r - a.min_count) <= span);
                if (lg == 0) tested_local += ok ? 1u : 0u;
                const float f = (float)n1r;
                const float sq = __builtin_amdgcn_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; 1 ulp sqrt; (1 - 2^-20)
                float2 tm;
                tm.x = ok ? sq : __builtin_huge_valf();
                tm.y = (a.eg_max + fminf(a.rall_max, f * a.rmax_max)) * 1.000001f;
                if (KGWAS_COARSE_ABLATE & 16) tm = make_float2(1000.0f + m, 1.0f);
                trm[lane] = tm;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < RT * 4; i += 2) {
                    const float4 v = *reinterpret_cast<const float4*>(trm + kg * 16u + i);
                    sqd[i] = v.x;
                    er[i] = v.y;
                    sqd[i + 1] = v.z;
                    er[i + 1] = v.w;
                }
                __builtin_amdgcn_wave_barrier();
            }
            TL_STAMP(51);  // row terms exchanged
            // The test, ~2.5 lane-ops per pair: margin = |Dc| - alpha_p * sqrt(d) (float(Dc), one fma with |.| and
            // negation as operand modifiers), a running maximum per row slot (max3), and ONE compare per row slot:
            // the slot has a survivor iff max margin + E >= 0. Which pairs passed is only worked out for row slots
            // with a hit (steady state: ~2 of the 1024 (lane, slot) units of a pass).
            auto pair_margin = [&](int i, int g, float al) {
                int dc = acc[i >> 2][NS * g][i & 3];
                if (NS == 2) dc = __mul24(dc, 254) + acc[i >> 2][NS * g + NS - 1][i & 3];  // |D0| <= 127 * S < 2^23
                return fmaf(-al, sqd[i], fabsf((float)dc));  // NaN (frozen column) never wins a maximum
            };
            float alc[PG];
#pragma unroll
            for (int g = 0; g < PG; g++) alc[g] = colc[g * 16 + m];
            uint64_t hit[RT * 4];
            if (!(KGWAS_COARSE_ABLATE & 1)) {
                float mx[RT * 4];
#pragma unroll
                for (int i = 0; i < RT * 4; i++) mx[i] = -__builtin_huge_valf();
#pragma unroll
                for (int g = 0; g < PG; g++)
#pragma unroll
                    for (int i = 0; i < RT * 4; i++) mx[i] = fmaxf(mx[i], pair_margin(i, g, alc[g]));
#pragma unroll
                for (int i = 0; i < RT * 4; i++) hit[i] = __ballot(mx[i] + er[i] >= 0.0f);
            } else {
#pragma unroll
                for (int i = 0; i < RT * 4; i++) hit[i] = 0;
                int x = 0;  // keeps every accumulator (and its MFMAs) alive at one lane-op each
#pragma unroll
                for (int i = 0; i < RT * 4; i++)
#pragma unroll
                    for (int t = 0; t < T; t++) x ^= acc[i >> 2][t][i & 3];
                hit[0] = __ballot(x == 0x7fffffff);
            }
            TL_STAMP(52);  // tests done
            uint64_t hit_any = 0;
#pragma unroll
            for (int i = 0; i < RT * 4; i++) hit_any |= hit[i];
            if ((KGWAS_COARSE_ABLATE & 32) && hit_any) {  // tests only: a token side effect instead of the emission
                if (lane == 0) atomicAdd(&a.tested[0], 0ull);
                hit_any = 0;
            }
            // mb[g] bit i = pair (row slot i, column g*16 + m) survives; rebuilt for the row slots that had a hit.
            uint32_t mb[PG];
#pragma unroll
            for (int g = 0; g < PG; g++) mb[g] = 0;
            if (hit_any) {
#pragma unroll
                for (int i = 0; i < RT * 4; i++) {
                    if (hit[i]) {  // wave-uniform
#pragma unroll
                        for (int g = 0; g < PG; g++) {
                            float al = alc[g];
                            asm volatile("" : "+v"(al));  // recompute here: keeping all 16 * PG compare masks alive costs more
                            mb[g] |= (pair_margin(i, g, al) + er[i] >= 0.0f) ? (1u << i) : 0u;
                        }
                    }
                }
            }
            if (hit_any) {  // wave-uniform
                // Column (g, m)'s 64 row bits of this pass sit in four lanes (kg = 0..3; bit i = 4 rt + jj of mb[g] is row
                // rt * 16 + 4 kg + jj). Each lane stores its 16 bits as quarter kg of the column's word: no cross-lane
                // traffic, and launch_bitmap_keys(nibble_transposed = true) puts the word's nibbles back in row order.
                unsigned short* bm16 = reinterpret_cast<unsigned short*>(a.bitmap) + (rbase >> 6) * 4u + kg;
#pragma unroll
                for (int g = 0; g < PG; g++)
                    if (mb[g]) bm16[(uint64_t)colp[g * 16 + m] * a.words_per_col * 4u] = (unsigned short)mb[g];  // column >= 0 wherever a bit is set
            }
            TL_STAMP(53);  // hits resolved, survivors emitted
        }
    }
    if (a.tested) {
        uint32_t v = tested_local;  // every lane counted one row per pass
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// ---- survivors -> exact candidates, compacted in (column, row) order --------------------------------------------
// The sorted key list is cut into tiles of 256 survivors that never straddle a column (launch_bitmap_keys fills tile_pref); the two
// kernels below walk the tiles with a fixed grid:
//   rescore_kernel      exact re-scoring of a tile's survivors (one lane each, the column wave-uniform), exact test
//                       against thr, threshold histogram; score (or -inf: not a candidate) to HBM, candidates per tile
//   compact_kernel      candidates to their final place: score f64 | kmer u64 | chunk-local row u32, three arrays (a tile's
//                       place = the candidates of the tiles before it, summed by the block itself; the grid's last block
//                       writes each column's range in the compacted list and their total: meta)
// The compacted arrays live in HBM; the host copies exactly `total` records per array over PCIe on a copy stream
// (the first version wrote every survivor, candidate or not, from the kernel into mapped host memory: 20 B per
// survivor over PCIe inside the compute stream, 7 ms of a 22 ms pass).
__device__ __forceinline__ uint32_t tile_column(const uint32_t* tile_pref, uint32_t n_pheno, uint32_t t) {
    uint32_t lo = 0, hi = n_pheno;  // last p with tile_pref[p] <= t
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_pref[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// Same arithmetic as score_valu_kernel: the reference's select-and-add chains, then the exact candidate test.
// One 128-sample block of the four float32 chains of calculate_kmer_score (src/kmer_general.cpp:155-167): lane l of the
// reference's SSE register adds `bit ? y : +0.0f` for samples 128 b + 32 l + 31 - s, s = 0..31.
// The column's values are the same for every lane: they are read through the constant address space, i.e. by scalar
// loads into SGPRs (yb is wave-uniform). As vector loads - 32 x 16 bytes per lane and block, every lane the same
// address - they kept the CU's 64 B/clk vector-memory return path busier than the lane-ops kept the SIMDs.
typedef __attribute__((address_space(4))) const float* kconst_f32p;
// YLDS: yb points into LDS (the tile's column staged there by rescore_kernel): one broadcast ds_read_b128 per sample brings
// the four chains' values. The scalar loads this replaces return out of order, so every use waits for ALL of them
// (s_waitcnt lgkmcnt(0), six times per block) and the K$ misses on 201 columns x 8 KB: the waves were parked 61 % of their
// cycles with the vector ALUs 58 % busy (SQ_WAIT_ANY / SQ_ACTIVE_INST_VALU); LDS reads return in order. Re-score + the small
// kernels 4.06 -> 3.75 ms per 100 M rows at 1024 x 101, all kernels 49.9 -> 49.1 at 2048 x 201. (Also tried on top: the fma's
// 0.0f / 1.0f factors from a bank-conflict-free table in LDS instead of byte expansion + v_cvt_f32_ubyteK - a third of the
// vector instructions - measured +-0: twice the LDS reads then saturate the LDS pipe)
template <bool YLDS, int NS = 1>
__device__ __forceinline__ void rescore_block(const uint32_t (&w)[NS][4], const float* yb, float (&acc)[NS][4]) {
    kconst_f32p yc = (kconst_f32p)yb;
    const float4* yl = reinterpret_cast<const float4*>(yb);
    (void)yc;
    (void)yl;
#if KGWAS_RESCORE_FMA
    // acc + (bit ? y : +0) == fma((float)bit, y, acc) for finite y (the filters only run on finite phenotypes): the bits of a
    // dword become bytes 0 / 1 eight at a time ((w >> j) & 0x01010101: two lane-ops per four bits), v_cvt_f32_ubyteK makes
    // 0.0f / 1.0f of one, and v_pk_fma_f32 advances two chains: 2.0 lane-ops per sample instead of 2.5.
    // NS = 2 (experiments, launch_rescore): the lane carries two survivors of the column; the sample's four values are read once
    // for both (one broadcast ds_read_b128 per sample and survivor is 32 KB per wave and 128-sample block through a 128 B/clk
    // LDS pipe, against 896 issue cycles of lane-ops per SIMD) - measured slower, see launch_rescore.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#if KGWAS_RESCORE_HALVES
    // chains 0, 1 over the block's 32 steps, then chains 2, 3: sixteen expanded dwords live instead of thirty-two
    if (NS == 1 && YLDS) {
        const float2* yh = reinterpret_cast<const float2*>(yb);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t eh[2][8];
#pragma unroll
            for (int l = 0; l < 2; l++)
#pragma unroll
                for (int j = 0; j < 8; j++) eh[l][j] = (w[0][2 * half + l] >> j) & 0x01010101u;
            f32x2 ah = {acc[0][2 * half], acc[0][2 * half + 1]};
#pragma unroll
            for (int s = 0; s < 32; s++) {
                const int b = 31 - s, j = b & 7, k = b >> 3;
                const float2 yv = yh[2 * s + half];
                const f32x2 yy = {yv.x, yv.y};
                auto byte_f32 = [&](uint32_t x) {
                    float f;
                    if (k == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(x));
                    if (k == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(x));
                    if (k == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(x));
                    if (k == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(x));
                    return f;
                };
                const f32x2 g = {byte_f32(eh[0][j]), byte_f32(eh[1][j])};
                ah = __builtin_elementwise_fma(g, yy, ah);
            }
            acc[0][2 * half] = ah.x;
            acc[0][2 * half + 1] = ah.y;
        }
        return;
    }
#endif
    uint32_t e[NS][4][8];
#pragma unroll
    for (int v = 0; v < NS; v++)
#pragma unroll
        for (int l = 0; l < 4; l++)
#pragma unroll
            for (int j = 0; j < 8; j++) e[v][l][j] = (w[v][l] >> j) & 0x01010101u;
    f32x2 a01[NS], a23[NS];
#pragma unroll
    for (int v = 0; v < NS; v++) {
        a01[v] = (f32x2){acc[v][0], acc[v][1]};
        a23[v] = (f32x2){acc[v][2], acc[v][3]};
    }
#pragma unroll
    for (int s = 0; s < 32; s++) {
        const int b = 31 - s, j = b & 7, k = b >> 3;
        f32x2 y01, y23;
        if (YLDS) {
            const float4 yv = yl[s];
            y01 = (f32x2){yv.x, yv.y};
            y23 = (f32x2){yv.z, yv.w};
        } else {
            y01 = (f32x2){yc[4 * s], yc[4 * s + 1]};
            y23 = (f32x2){yc[4 * s + 2], yc[4 * s + 3]};
        }
        auto byte_f32 = [&](uint32_t x) {  // v_cvt_f32_ubyteK (the compiler shifts first and converts byte 0: two more ops per bit)
            float f;
            if (k == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(x));
            if (k == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(x));
            if (k == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(x));
            if (k == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(x));
            return f;
        };
#pragma unroll
        for (int v = 0; v < NS; v++) {
            const f32x2 g01 = {byte_f32(e[v][0][j]), byte_f32(e[v][1][j])};
            const f32x2 g23 = {byte_f32(e[v][2][j]), byte_f32(e[v][3][j])};
            a01[v] = __builtin_elementwise_fma(g01, y01, a01[v]);
            a23[v] = __builtin_elementwise_fma(g23, y23, a23[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < NS; v++) {
        acc[v][0] = a01[v].x;
        acc[v][1] = a01[v].y;
        acc[v][2] = a23[v].x;
        acc[v][3] = a23[v].y;
    }
    return;
#endif
#pragma unroll
    for (int s = 0; s < 32; s++) {
        const float yv[4] = {YLDS ? yb[4 * s] : yc[4 * s], YLDS ? yb[4 * s + 1] : yc[4 * s + 1], YLDS ? yb[4 * s + 2] : yc[4 * s + 2],
                             YLDS ? yb[4 * s + 3] : yc[4 * s + 3]};
#pragma unroll
        for (int v = 0; v < NS; v++)
#pragma unroll
            for (int l = 0; l < 4; l++) {
                // bit 31 - s as 0 / -1 in one v_bfe_i32, kept from the optimiser, which turns bfe & y (like a shift pair)
                // into and + compare + select: 3.5 lane-ops per sample, 2.5 this way with the packed add
                int mk = __builtin_amdgcn_sbfe((int)w[v][l], 31 - s, 1);
                asm volatile("" : "+v"(mk));
                acc[v][l] = acc[v][l] + __int_as_float(mk & __float_as_int(yv[l]));
            }
    }
}

// exact score of one survivor from its chain sums -> tmp_score (or -inf), and the tile's candidate count
__device__ __forceinline__ void rescore_finish(const ScoreArgs& a, uint32_t p, bool valid, uint32_t gi, const float (&acc)[4], uint32_t n1,
                                               double* tmp_score, uint32_t* tile_cnt, uint32_t t, uint32_t* wcnt) {
    const float yf = ((acc[0] + acc[1]) + acc[2]) + acc[3];
    double q, d, sc, out = -__builtin_huge_val();
    score_terms(a, yf, n1, a.sums[p], q, d);
    if (valid && mac_pass(a, n1) && candidate_score(a, p, q, d, a.thr[p], sc)) out = sc;
    if (valid) tmp_score[gi] = out;
    const uint32_t c = __popcll(__ballot(out != -__builtin_huge_val()));
    if ((threadIdx.x & 63u) == 0u) wcnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[t] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
}

// Rows are read in place by their own lane, 8 bytes x 2 per 128-sample block. (Staging a wave's 64 rows through LDS with
// coalesced copies - every line fetched once - measured 30-50 % SLOWER: the gathers are not what this kernel waits for.)
// Round 6 went through the remaining suspects with counters and build variants (tools/pmc_rescore_mem.sh, tools/coarse_variants.sh;
// re-score + small kernels per 100 M rows x 1024 x 101, alternating runs, 3.64-3.74 ms as it is): the counters of its large
// launches say 65 % of the wave cycles are spent in s_waitcnt and two thirds of the L1's accesses go on to the L2 (a row's line
// is asked for 16 times, a 128-sample block apart, and the L1 sees 1300 other rows in between) - and yet: ONE 16-byte request per
// block instead of two 8-byte ones 3.62-3.64; the wave fetching its 64 rows together, eight lanes per row, through a
// conflict-free LDS area (every line requested once) 3.77; all of a row's requests at once 5.6; a wave's candidates added to the
// column's histogram with one atomic per distinct bin instead of one each 4.44 (the loop over ~30 distinct bins costs more than
// the atomics did); a block taking a RUN of consecutive tiles, the column copied into LDS once per column instead of once per
// tile, 3.64. tools/probe_gather.hip: a bare gather of 2^20 rows in this access pattern takes 45 us while the rows sit in the L2
// and 135 us once they are spread over more than 256 MB - the kernel's launches over the chunks of 1-8 M rows are where its time
// goes (0.4 ms each for ~10^6 survivors against 0.09 ms for the same number in a chunk of 50 k rows). Left as it is.
// DIRECT (scans of a few columns, launch_rescore_direct): the survivor's record is written at its place in the key order -
// score or -inf, k-mer, row - and nothing is counted or compacted afterwards.
// NS = 2 (experiments): a block takes two consecutive tiles at a time and, where both belong to the same column - all but
// one pair per column -, every lane carries one survivor of each: the column's values are read from LDS once for both.
#ifndef KGWAS_RESCORE_WAVES
// experiments: waves per SIMD the register allocation is held to (0: the compiler's choice, 84 registers = 5 waves). Re-score +
// small kernels per 100 M rows x 1024 x 101: 3.66-3.70 ms as is; 6 waves + KGWAS_RESCORE_HALVES 3.80, 7 waves 3.76, 7 + halves
// 3.81-3.85, 8 + halves 4.13 (a few spills each), halves alone 3.89: neither more waves nor fewer LDS reads (KGWAS_RESCORE_NS=2)
// is what this kernel lacks.
#define KGWAS_RESCORE_WAVES 0
#endif
#if KGWAS_RESCORE_WAVES
#define KGWAS_RESCORE_OCC __attribute__((amdgpu_waves_per_eu(KGWAS_RESCORE_WAVES, KGWAS_RESCORE_WAVES)))
#else
#define KGWAS_RESCORE_OCC
#endif
template <bool YLDS, bool DIRECT = false, int NS = 1>
__global__ void __launch_bounds__(256) KGWAS_RESCORE_OCC rescore_kernel(ScoreArgs a, const uint32_t* keys, const uint32_t* surv_off,
                                                      const uint32_t* surv_cnt, const uint32_t* tile_pref, uint32_t row_mask,
                                                      double* tmp_score, uint32_t* tile_cnt, uint32_t* ticket) {
    __shared__ uint32_t wcnt[4];
    __shared__ uint32_t next_tile;
    extern __shared__ float4 ytile[];  // YLDS: the tile's column, 64 * W_m floats

    const uint32_t n_tiles = tile_pref[a.n_pheno];
    const uint32_t L = 64u * a.W_m;
    const uint32_t nblk = a.W_m / 2u;
    // tiles t0 .. t0 + nv - 1 of column p (nv <= NS)
    auto run = [&](uint32_t t0, uint32_t p, uint32_t nv) {
        bool valid[NS];
        uint32_t gi[NS];
        uint64_t r[NS];
        const uint32_t* rp[NS];
#pragma unroll
        for (int v = 0; v < NS; v++) {
            const uint32_t i = (t0 + (uint32_t)v - tile_pref[p]) * 256u + threadIdx.x;
            valid[v] = (uint32_t)v < nv && i < surv_cnt[p];
            gi[v] = surv_off[p] + (valid[v] ? i : 0u);
            r[v] = keys[gi[v]] & row_mask;
            rp[v] = a.src.base + r[v] * a.src.stride_dw + a.src.off_dw;
        }
        float acc[NS][4];
        uint32_t n1[NS];
#pragma unroll
        for (int v = 0; v < NS; v++) {
            n1[v] = 0;
#pragma unroll
            for (int l = 0; l < 4; l++) acc[v][l] = 0.0f;
        }
        if (YLDS) {  // (the previous tile's readers are past rescore_finish's barriers)
            const float4* ysrc = reinterpret_cast<const float4*>(a.Yperm + (size_t)p * L);
            for (uint32_t k = threadIdx.x; k < L / 4u; k += 256u) ytile[k] = ysrc[k];
            __syncthreads();
        }
        // the rows' words for block b + 1 are asked for before block b's lane-ops start (round 6: ALL of a row's words - eight
        // blocks at 1024 samples - requested at once, before the first block: re-score + small kernels 5.6 ms per 100 M rows x
        // 1024 x 101 against 3.67, four blocks at a time 4.45: the wave-loads, 64 cache lines each, then queue up in the
        // texture-address path instead of being spread between the lane-ops)
        uint2 nx[NS][2];
        auto fetch = [&](uint32_t b) {
#pragma unroll
            for (int v = 0; v < NS; v++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    nx[v][h] = make_uint2(0u, 0u);
                    if (4u * b + 2u * h + 1u < a.src.avail_dw) nx[v][h] = *reinterpret_cast<const uint2*>(rp[v] + 4u * b + 2u * h);
                }
        };
        if (nblk) fetch(0);
        for (uint32_t b = 0; b < nblk; b++) {
            uint32_t w[NS][4];
#pragma unroll
            for (int v = 0; v < NS; v++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    w[v][2 * h] = nx[v][h].x & a.dmask[4 * b + 2 * h];
                    w[v][2 * h + 1] = nx[v][h].y & a.dmask[4 * b + 2 * h + 1];
                }
                n1[v] += __popc(w[v][0]) + __popc(w[v][1]) + __popc(w[v][2]) + __popc(w[v][3]);
            }
            if (b + 1u < nblk) fetch(b + 1u);
            if (YLDS)
                rescore_block<true, NS>(w, reinterpret_cast<const float*>(ytile) + 128u * b, acc);
            else
                rescore_block<false, NS>(w, a.Yperm + (size_t)p * L + 128u * b, acc);
        }
        if (DIRECT) {
#pragma unroll
            for (int v = 0; v < NS; v++) {
                const float yf = ((acc[v][0] + acc[v][1]) + acc[v][2]) + acc[v][3];
                double q, d, sc, out = -__builtin_huge_val();
                score_terms(a, yf, n1[v], a.sums[p], q, d);
                if (valid[v] && mac_pass(a, n1[v]) && candidate_score(a, p, q, d, a.thr[p], sc)) out = sc;
                if (valid[v]) {
                    a.so_score[gi[v]] = out;
                    a.so_kmer[gi[v]] = a.file_rows[r[v] * a.file_stride_w];
                    a.so_row[gi[v]] = (uint32_t)r[v];
                }
            }
            if (YLDS) __syncthreads();  // the next tile's column overwrites ytile
        } else {
#pragma unroll
            for (int v = 0; v < NS; v++)
                if ((uint32_t)v < nv) rescore_finish(a, p, valid[v], gi[v], acc[v], n1[v], tmp_score, tile_cnt, t0 + (uint32_t)v, wcnt);  // (nv is block-uniform)
        }
    };
    // (tiles by ticket where the caller passes one - the counter is zeroed by the chunk's prep launch -: experiments, launch_rescore)
    for (uint32_t t = (uint32_t)NS * blockIdx.x;; t += (uint32_t)NS * gridDim.x) {
        if (ticket) {
            if (threadIdx.x == 0) next_tile = atomicAdd(ticket, (uint32_t)NS);
            __syncthreads();
            t = next_tile;
            __syncthreads();
        }
        if (t >= n_tiles) break;
        const uint32_t p = __builtin_amdgcn_readfirstlane(tile_column(tile_pref, a.n_pheno, t));
        if (NS == 1) {
            run(t, p, 1u);
        } else {
            const bool two = t + 1u < n_tiles && t + 1u < tile_pref[p + 1u];  // the next tile is the same column's
            run(t, p, two ? 2u : 1u);
            if (!two && t + 1u < n_tiles) run(t + 1u, __builtin_amdgcn_readfirstlane(tile_column(tile_pref, a.n_pheno, t + 1u)), 1u);
        }
    }
}

// Candidates to their final place. A tile's place in the compacted arrays is the number of candidates in the tiles before
// it: every block adds those up for itself - its first tile's prefix over tile_cnt[0 .. t), then gridDim.x - 1 more counts
// per further tile - instead of waiting for a one-block scan launch between the re-score and this kernel (16 us per chunk,
// 0.4 ms of a 100 M-row pass). The grid's extra last block writes meta: [0 .. P) candidates per column, [P .. 2P) each
// column's offset in the compacted arrays, [2P] their total, [2P + 1] the survivor keys the filter emitted (> key_cap: the
// list overflowed).
__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t* red) {  // sum over the block's 256 threads; two barriers
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    __syncthreads();  // (the previous call's readers are done with red)
    if (lane == 0u) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) compact_kernel(ScoreArgs a, const uint32_t* keys, const uint32_t* surv_off,
                                                      const uint32_t* surv_cnt, const uint32_t* tile_pref, uint32_t row_mask,
                                                      const double* tmp_score, const uint32_t* tile_cnt, const uint32_t* key_count,
                                                      uint32_t* meta) {
    __shared__ uint32_t wcnt[4];
    __shared__ uint32_t red[4];
    __shared__ uint32_t cpre[257];
    const uint32_t n_tiles = tile_pref[a.n_pheno];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_workers = gridDim.x - 1u;
    if (blockIdx.x == n_workers) {
        // meta. Thread i sums the counts of tiles [i * per, (i + 1) * per); an exclusive scan over the 256 sums; a column's
        // offset is the prefix at its first tile.
        const uint32_t per = (n_tiles + 255u) / 256u;
        const uint32_t lo = threadIdx.x * per < n_tiles ? threadIdx.x * per : n_tiles;
        const uint32_t hi = lo + per < n_tiles ? lo + per : n_tiles;
        uint32_t sm = 0;
        for (uint32_t i = lo; i < hi; i++) sm += tile_cnt[i];
        cpre[threadIdx.x + 1u] = sm;
        if (threadIdx.x == 0u) cpre[0] = 0u;
        __syncthreads();
        for (uint32_t d = 1; d < 256u; d <<= 1) {
            const uint32_t x = threadIdx.x + 1u > d ? cpre[threadIdx.x + 1u - d] : 0u;  // (cpre[0] = 0 joins nothing)
            __syncthreads();
            cpre[threadIdx.x + 1u] += x;
            __syncthreads();
        }
        auto prefix_at = [&](uint32_t x) {  // candidates in tiles [0, x)
            if (x >= n_tiles) return cpre[256];
            const uint32_t seg = per ? x / per : 0u;
            uint32_t v = cpre[seg];
            for (uint32_t i = seg * per; i < x; i++) v += tile_cnt[i];
            return v;
        };
        for (uint32_t p = threadIdx.x; p < a.n_pheno; p += 256u) {
            const uint32_t o0 = prefix_at(tile_pref[p]), o1 = prefix_at(tile_pref[p + 1u]);
            meta[p] = o1 - o0;
            meta[a.n_pheno + p] = o0;
        }
        if (threadIdx.x == 0u) {
            meta[2u * a.n_pheno] = cpre[256];
            meta[2u * a.n_pheno + 1u] = *key_count;
        }
        return;
    }
    uint32_t base = 0, done = 0;  // candidates in tiles [0, done)
    for (uint32_t t = blockIdx.x; t < n_tiles; t += n_workers) {
        uint32_t part = 0;
        for (uint32_t i = done + threadIdx.x; i < t; i += 256u) part += tile_cnt[i];
        base += block_sum_256(part, red);
        done = t;
        if (tile_cnt[t] == 0u) continue;  // block-uniform: no candidate in this tile
        const uint32_t p = __builtin_amdgcn_readfirstlane(tile_column(tile_pref, a.n_pheno, t));  // the column's values come by scalar loads
        const uint32_t i = (t - tile_pref[p]) * 256u + threadIdx.x;
        const bool valid = i < surv_cnt[p];
        const uint32_t gi = surv_off[p] + (valid ? i : 0u);
        const double sc = valid ? tmp_score[gi] : -__builtin_huge_val();
        const bool is_cand = sc != -__builtin_huge_val();
        const unsigned long long bal = __ballot(is_cand);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; w++) before += wcnt[w];
        if (is_cand) {
            const uint32_t r = keys[gi] & row_mask;
            const uint32_t o = base + before;
            a.so_score[o] = sc;
            a.so_kmer[o] = a.file_rows[(uint64_t)r * a.file_stride_w];
            a.so_row[o] = r;
        }
        __syncthreads();
    }
}

// One launch instead of a handful of small memsets per chunk: counters of the next chunk, its survivors' bitmap, the narrow
// filter's per-segment counts.
__global__ void __launch_bounds__(256) chunk_prep_kernel(uint32_t* cand_cnt, uint32_t n_pheno, unsigned long long* tested, uint32_t* key_count,
                                                         unsigned long long* bitmap, uint64_t bitmap_words, uint32_t* seg_cnt, uint32_t n_seg_words,
                                                         PrepThr tu) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pheno) cand_cnt[i] = 0u;
    if (i < TESTED_SHARDS) tested[i] = 0ull;
    if (i == 0 && key_count) key_count[0] = 0u, key_count[1] = 0u;  // ([1]: rescore_kernel's tile ticket)
    if (i < n_seg_words) seg_cnt[i] = 0u;
    // the bitmap, two words (16 bytes) per thread and turn
    ulonglong2* b2 = reinterpret_cast<ulonglong2*>(bitmap);
    const uint64_t pairs = bitmap_words / 2u;
    for (uint64_t k = i; k < pairs; k += (uint64_t)gridDim.x * blockDim.x) b2[k] = make_ulonglong2(0ull, 0ull);
    if (i == 0 && (bitmap_words & 1u)) bitmap[bitmap_words - 1u] = 0ull;
    // the thresholds the chunk is filtered against, raised from the histograms of everything counted so far
    if (tu.hist)
        for (uint32_t p = blockIdx.x; p < n_pheno; p += gridDim.x) thr_update_block(tu.hist, tu.hist_base, tu.bins, tu.topn, tu.thr_host, tu.thr, p);
}

hipError_t launch_chunk_prep(uint32_t* cand_cnt, uint32_t n_pheno, unsigned long long* tested, uint32_t* key_count,
                             unsigned long long* bitmap, uint64_t bitmap_words, uint32_t* seg_cnt, uint32_t n_seg_words, const PrepThr& tu,
                             hipStream_t st) {
    uint64_t n = n_pheno > TESTED_SHARDS ? n_pheno : TESTED_SHARDS;
    if (n_seg_words > n) n = n_seg_words;
    if (bitmap_words / 8u > n) n = bitmap_words / 8u;  // (four turns per thread on a large bitmap)
    const uint64_t blocks = std::min<uint64_t>((n + 255u) / 256u, 8192u);
    if (blocks * 256u < std::max<uint64_t>(std::max<uint64_t>(n_pheno, TESTED_SHARDS), n_seg_words)) return hipErrorInvalidValue;
    if (tu.hist && (tu.bins % 1024u || tu.bins / 256u > 64u)) return hipErrorInvalidValue;
    launch_last(chunk_prep_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, cand_cnt, n_pheno, tested, key_count, bitmap, bitmap_words,
                       seg_cnt, n_seg_words, tu);
    return hipGetLastError();
}

// B operands of one LDS group + the group's per-column constants (3 x up to 128 words) + the eight waves' row-term
// exchange areas
size_t coarse_lds_bytes(uint32_t n_kgroups, uint32_t T) { return (size_t)n_kgroups * 8u * T * 1024u + 1536u + 8u * 768u; }

template <int T, int NS, int TH = 512>
static hipError_t launch_coarse_t(const CoarseArgs& a, uint32_t rows_per_block, uint32_t n_rowblocks, size_t lds,
                                  hipStream_t st) {
    lds += (size_t)(TH / 64 - 8) * 768u;  // per-wave scratch of the waves beyond eight
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)coarse_kernel<T, NS, TH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    static const int grid_env = (int)exp_int("KGWAS_COARSE_GRIDLG", -1);  // experiments
    const uint32_t grid_lg = (grid_env >= 0 ? grid_env != 0 : true) && a.n_lgroups > 1 ? 1u : 0u;
    const uint32_t grid = grid_lg ? (n_rowblocks + 7u) / 8u * 8u * a.n_lgroups : n_rowblocks;
    launch_last(coarse_kernel<T, NS, TH>, dim3(grid), dim3(TH), lds, st, a, rows_per_block, n_rowblocks, grid_lg);
    return hipGetLastError();
}

hipError_t launch_coarse(const CoarseArgs& a, uint32_t T, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const size_t lds = coarse_lds_bytes(a.n_kgroups, T);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    // Few tiles per LDS group leave registers for a third wave per SIMD (T <= 4: <= 168 VGPRs at 768 threads)
    static const int th_env = (int)exp_int("KGWAS_COARSE_TH", 768);  // experiments (512: eight waves as for T > 4)
    const bool wide_block = th_env == 768 && a.n_slices == 1 && T <= 4 && lds + 4u * 768u <= 160u * 1024u;
    const uint32_t threads = wide_block ? 768 : 512;
    if ((a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit byte offsets
    const uint32_t rpp = (threads >> 6) * 64u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
#ifdef KGWAS_COARSE_BENCH_ONLY  // experiments: only the two shapes of the 1024 x 101 bench (fast compiles)
    if (a.n_slices == 1 && T == 7) return launch_coarse_t<7, 1>(a, rows_per_block, n_rowblocks, lds, st);
    if (a.n_slices == 2 && T == 8) return launch_coarse_t<8, 2>(a, rows_per_block, n_rowblocks, lds, st);
    return hipErrorInvalidValue;
#else
    if (wide_block) {
        switch (T) {
            case 1: return launch_coarse_t<1, 1, 768>(a, rows_per_block, n_rowblocks, lds, st);
            case 2: return launch_coarse_t<2, 1, 768>(a, rows_per_block, n_rowblocks, lds, st);
            case 3: return launch_coarse_t<3, 1, 768>(a, rows_per_block, n_rowblocks, lds, st);
            case 4: return launch_coarse_t<4, 1, 768>(a, rows_per_block, n_rowblocks, lds, st);
        }
    }
    if (a.n_slices == 1) {
        switch (T) {
            case 1: return launch_coarse_t<1, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 2: return launch_coarse_t<2, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 3: return launch_coarse_t<3, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 4: return launch_coarse_t<4, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 5: return launch_coarse_t<5, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 6: return launch_coarse_t<6, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 7: return launch_coarse_t<7, 1>(a, rows_per_block, n_rowblocks, lds, st);
            case 8: return launch_coarse_t<8, 1>(a, rows_per_block, n_rowblocks, lds, st);
        }
    } else if (a.n_slices == 2) {
        switch (T) {
            case 2: return launch_coarse_t<2, 2>(a, rows_per_block, n_rowblocks, lds, st);
            case 4: return launch_coarse_t<4, 2>(a, rows_per_block, n_rowblocks, lds, st);
            case 6: return launch_coarse_t<6, 2>(a, rows_per_block, n_rowblocks, lds, st);
            case 8: return launch_coarse_t<8, 2>(a, rows_per_block, n_rowblocks, lds, st);
        }
    }
    return hipErrorInvalidValue;
#endif
}

#ifdef KGWAS_COARSE_TIMELINE
extern "C" int kgwas_debug_coarse_timeline(unsigned long long* out, unsigned long long n) {
    if (n > 2 * 16 * 64) n = 2 * 16 * 64;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coarse_tl), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

hipError_t launch_rescore(const ScoreArgs& a, const uint32_t* keys, const uint32_t* surv_off, const uint32_t* surv_cnt,
                          uint32_t row_bits, uint32_t* tile_pref, uint32_t* tile_cnt, double* tmp_score,
                          const uint32_t* key_count, uint32_t* meta, hipStream_t st) {
    if (a.n_pheno == 0) return hipSuccess;
    const uint32_t row_mask = row_bits >= 32 ? 0xFFFFFFFFu : ((1u << row_bits) - 1u);
    // fixed grid, tiles handed out round-robin: 8 blocks of 256 per CU (the tile count is only known on the device)
    // the tile's column goes through LDS while it fits beside eight blocks per CU (16 KB: 4096 samples); beyond, scalar loads
    const size_t ybytes = 64u * (size_t)a.W_m * sizeof(float);
    static const bool ylds_ok = exp_str("KGWAS_RESCORE_YLDS") ? atoi(exp_str("KGWAS_RESCORE_YLDS")) != 0 : true;  // experiments
    // (KGWAS_RESCORE_NS=2, experiments: two survivors per lane, the column's values read from LDS once for both - half the LDS
    // reads, but 146 registers instead of 84, three waves per SIMD instead of five: all scoring kernels 14.0 ms per 100 M rows x
    // 1024 x 101 against 13.5; what this kernel waits for is its row gathers, and fewer waves hide them worse)
    [[maybe_unused]] static const int ns_env = (int)exp_int("KGWAS_RESCORE_NS", 1);
    // (KGWAS_RESCORE_DYN=1, experiments: tiles handed out by an atomic ticket instead of round-robin - a launch is 800-3700 tiles on
    // 1280 block slots - measured slower: re-score + small kernels 4.10-4.12 ms per 100 M rows x 1024 x 101 against 3.65-3.71)
    static const bool dyn = exp_int("KGWAS_RESCORE_DYN", 0) != 0;
    static const uint32_t grid = (uint64_t)exp_int("KGWAS_RESCORE_GRID", 2048u);
    uint32_t* ticket = dyn ? const_cast<uint32_t*>(key_count) + 1 : nullptr;  // (d_key_count has two words; chunk_prep_kernel zeroes both)
#ifdef KGWAS_EXPERIMENTS
    if (ylds_ok && ybytes <= 16384u && ns_env == 2)
        hipLaunchKernelGGL((rescore_kernel<true, false, 2>), dim3(grid), dim3(256), ybytes, st, a, keys, surv_off, surv_cnt, tile_pref, row_mask,
                           tmp_score, tile_cnt, ticket);
    else
#endif
    if (ylds_ok && ybytes <= 16384u)
        hipLaunchKernelGGL(rescore_kernel<true>, dim3(grid), dim3(256), ybytes, st, a, keys, surv_off, surv_cnt, tile_pref, row_mask,
                           tmp_score, tile_cnt, ticket);
    else
        hipLaunchKernelGGL(rescore_kernel<false>, dim3(grid), dim3(256), 0, st, a, keys, surv_off, surv_cnt, tile_pref, row_mask,
                           tmp_score, tile_cnt, ticket);
    launch_last(compact_kernel, dim3(2048 + 1), dim3(256), 0, st, a, keys, surv_off, surv_cnt, tile_pref, row_mask, tmp_score,
                       tile_cnt, key_count, meta);
    return hipGetLastError();
}

hipError_t launch_rescore_direct(const ScoreArgs& a, const uint32_t* keys, const uint32_t* surv_off, const uint32_t* surv_cnt,
                                 uint32_t row_bits, const uint32_t* tile_pref, hipStream_t st) {
    if (a.n_pheno == 0) return hipSuccess;
    const uint32_t row_mask = row_bits >= 32 ? 0xFFFFFFFFu : ((1u << row_bits) - 1u);
    const size_t ybytes = 64u * (size_t)a.W_m * sizeof(float);
    // (a few tiles per launch: a small grid - an empty block costs its launch slot, and a scan of a few columns is made of
    // these launches)
    if (ybytes <= 16384u)
        launch_last(rescore_kernel<true, true>, dim3(512), dim3(256), ybytes, st, a, keys, surv_off, surv_cnt, tile_pref, row_mask,
                           (double*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    else
        launch_last(rescore_kernel<false, true>, dim3(512), dim3(256), 0, st, a, keys, surv_off, surv_cnt, tile_pref, row_mask,
                           (double*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    return hipGetLastError();
}

}  // namespace kgwas
