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
// the way that needs no data exchange: in a 512-sample group g, lane (m = lane&15, kg = lane>>4) loads ITS
// OWN 16 bytes of row m (samples 512g+128kg .. +127) and MFMA j = 0..7 uses bits 16j..16j+15 of them, i.e.
//        k = 16*kg + e   <->   sample 512g + 128kg + 16j + e.
// The four kg-lanes of a row thus fetch 64 contiguous bytes per instruction and every byte is used by the
// lane that loaded it. Bits become int8 0/1 with (nibble * 0x00204081) & 0x01010101 (4 bytes per 3 VALU ops);
// each expanded operand feeds T column tiles, each B operand (ds_read_b128) four row tiles.
#include <algorithm>

#include "score_common.h"

#ifndef KGWAS_COARSE_PF
#define KGWAS_COARSE_PF 0  // 0: load each unit when it starts, 1: prefetch the next sample group, 2: also across passes
#endif                     // (measured: 12.3 / 12.5 / 13.4 ms per pass of one-slice launches; 1 and 2 cost registers)
#define COARSE_SBUF 2048u  // survivor keys a block buffers in LDS
#ifndef KGWAS_COARSE_PHASES
#define KGWAS_COARSE_PHASES 1  // pin the per-step order: LDS reads, operand expansion, MFMAs
#endif
#ifndef KGWAS_COARSE_ABLATE
#define KGWAS_COARSE_ABLATE 0  // timing experiments only (wrong results): 1 no test, 2 no expansion, 4 no LDS reads, 8 no row loads
#endif

namespace kgwas {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// 16 presence bits -> 16 int8 0/1 (k-element e in byte e).
__device__ __forceinline__ i32x4 expand16(uint32_t x) {
    i32x4 r;
    r[0] = (int)(((x & 0xFu) * 0x00204081u) & 0x01010101u);
    r[1] = (int)((((x >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
    r[2] = (int)((((x >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
    r[3] = (int)((((x >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
    return r;
}

// T  = operand tiles (16 columns of one int8 slice each) per LDS group,
// NS = int8 slices per column (1: tile t of a group = column tile; 2: tiles 2g, 2g+1 = the two slices of column group g).
//
// The conservative test. The host centres the quantisation at c = sum/N (sum = the reference's float32 sum), so
//   r_c = N*yc - N1*sum = N*u*Dc            Dc = D0 (one slice) or 254*D0 + D1 (two slices, s0 = 254*s1), u = s0 or s1,
// an exact integer times a column constant. A pair can only have score_ref > thr if
//   |Dc| >= alpha_p * sqrt(d(N1)) - E(N1)/u_p,    alpha_p = sqrt(thr_p)/(N*u_p),  E(N1) = Eg + min(Rall, N1*rmax)
// with Eg, Rall, rmax (header) taken as their maxima over all columns, so E is a per-ROW term (permutation columns
// share them anyway). The right-hand side is evaluated in float32 from constants the host rounded in the safe
// direction (alpha down by 2^-19 relative, sqrt(d) down by 2^-20, E and 1/u up by 1e-6), which dominates the
// float32 rounding of the three operations: the computed limit never exceeds the true one, so no possible
// candidate is dropped. 5 cheap lane-ops per pair instead of 14 double-precision ones.
//
// N1 comes out of the matrix pipe as well: the last operand column of every LDS group holds 1 for each phenotyped
// sample (0 elsewhere), so its dot product IS the masked popcount - no masks, no popcounts, no mask loads in the
// main loop, and bits of unphenotyped samples or row padding meet zeros in every operand column.
template <int T, int NS>
__global__ void __launch_bounds__(512) coarse_kernel(CoarseArgs a, uint32_t rows_per_block, uint32_t n_rowblocks) {
    extern __shared__ i32x4 blds[];  // [n_kgroups][8][T][64] x 16 bytes, colc[3][PG*16] (alpha, 1/u, column index), survivor buffer
    constexpr int PG = T / NS;       // column groups (of 16) per LDS group
    constexpr int RT = 4;
    constexpr int SLOTS = PG * 16;
    const uint32_t rb = blockIdx.x;
    if (rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t kg = lane >> 4, m = lane & 15u;
    const uint32_t n_kgroups = a.n_kgroups;
    const uint32_t group_vec = n_kgroups * 8u * T * 64u;  // i32x4 elements per LDS group
    float* colc = reinterpret_cast<float*>(blds + group_vec);
    const int* colp = reinterpret_cast<const int*>(colc + 2 * SLOTS);
    // Survivors are keys (column << row_bits | chunk-local row). A block collects them in LDS and appends them to the
    // global list with ONE atomic at its end (per-column counters bumped once per wave pass were ~9e5 device-scope
    // atomics per launch on ~100 addresses: they serialise at the memory side and cost more than the MFMAs).
    uint32_t* sctl = reinterpret_cast<uint32_t*>(colc + 3 * SLOTS);  // [0] reserved, [1] start of the first reservation that did not fit, [2] global base, [3] n
    uint32_t* sbuf = sctl + 4;
    if (threadIdx.x == 0) {
        sctl[0] = 0u;
        sctl[1] = 0xFFFFFFFFu;
    }
    const uint32_t rows_per_pass = (blockDim.x >> 6) * (RT * 16u);
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    uint32_t tested_local = 0;

    for (uint32_t lg = 0; lg < a.n_lgroups; lg++) {
        if (lg) __syncthreads();
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
        // (bytes 64g + 16kg .. +15 of the row's bits; zero beyond its data). The pieces of the NEXT unit - the next
        // sample group, or group 0 of the wave's next 64 rows - are fetched into the same registers as soon as the
        // current unit has expanded them (first half after step 3, second half after step 7): ~4 steps (2000 cycles)
        // ahead of their use, and across the epilogue, which two waves per SIMD could not hide otherwise.
        uint32_t piece[RT][4];
        uint32_t ro[RT];  // 32-bit byte offsets (launch_coarse guarantees the chunk spans < 4 GiB) of the rows to fetch next
        auto set_rows = [&](uint64_t rb0) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint64_t r = rb0 + rt * 16u + m;
                if (r >= a.n_rows) r = a.n_rows - 1;
                ro[rt] = ((uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw + 4u * kg) * 4u;
            }
        };
        auto load_half = [&](uint32_t g, int h) {
            const uint32_t b0 = 64u * g + 8u * h;  // + 16*kg is part of ro
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint2 v = make_uint2(0u, 0u);
                if ((KGWAS_COARSE_ABLATE & 8) == 0 && b0 + 16u * kg + 8u <= avail_b)
                    v = *reinterpret_cast<const uint2*>(rows_base + (ro[rt] + b0));
                if (KGWAS_COARSE_ABLATE & 8) v = make_uint2(ro[rt] + b0, lane * 2654435761u);
                piece[rt][2 * h] = v.x;
                piece[rt][2 * h + 1] = v.y;
            }
        };
        const uint64_t wave_row0 = blk_row0 + wave * (RT * 16u);
        if (KGWAS_COARSE_PF == 2 && wave_row0 < a.n_rows) {
            set_rows(wave_row0);
            load_half(0, 0);
            load_half(0, 1);
        }

        for (uint32_t ps = 0; ps * rows_per_pass < rows_per_block; ps++) {
            const uint64_t rbase = wave_row0 + (uint64_t)ps * rows_per_pass;
            if (rbase >= a.n_rows) break;  // wave-uniform
            const uint64_t rnext = rbase + rows_per_pass;
            const bool next_pass = (KGWAS_COARSE_PF == 2) && ((ps + 1) * rows_per_pass < rows_per_block) && (rnext < a.n_rows);  // wave-uniform
            if (KGWAS_COARSE_PF < 2) {
                set_rows(rbase);
                load_half(0, 0);
                load_half(0, 1);
            }
            i32x4 acc[RT][T];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int t = 0; t < T; t++) acc[rt][t] = (i32x4){0, 0, 0, 0};

            for (uint32_t g = 0; g < n_kgroups; g++) {
                const bool last_g = g + 1 == n_kgroups;
                const bool fetch = (KGWAS_COARSE_PF >= 1) && (!last_g || next_pass);  // is there a next unit (wave-uniform)
                const uint32_t gn = last_g ? 0u : g + 1u;
                const i32x4* bg = blds + (size_t)g * 8u * T * 64u + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (KGWAS_COARSE_PF == 0 && j == 0 && g > 0) {
                        load_half(g, 0);
                        load_half(g, 1);
                    }
                    // Phase order per step: all B operands of the step are requested from LDS first, the A operands are
                    // expanded while those reads are in flight, then the T x 4 MFMAs issue back to back (the wave's
                    // partner on the SIMD runs its own load/expand phase underneath them).
                    i32x4 B[T];
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        if (KGWAS_COARSE_ABLATE & 4)
                            B[t] = (i32x4){(int)lane, t, j, (int)g};
                        else
                            B[t] = bg[(j * T + t) * 64];
                    }
#if KGWAS_COARSE_PHASES
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    i32x4 A[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        if (KGWAS_COARSE_ABLATE & 2)
                            A[rt] = (i32x4){(int)piece[rt][0], (int)piece[rt][1], (int)piece[rt][2], (int)piece[rt][3] + j};
                        else
                            A[rt] = expand16((piece[rt][j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
                    }
                    if (j == 3 && fetch) {  // dwords 0,1 served j = 0..3
                        if (last_g) set_rows(rnext);
                        load_half(gn, 0);
                    }
                    if (j == 7 && fetch) load_half(gn, 1);  // dwords 2,3 served j = 4..7
#if KGWAS_COARSE_PHASES
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int t = 0; t < T; t++) {
#pragma unroll
                        for (int rt = 0; rt < RT; rt++)
                            acc[rt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt], B[t], acc[rt][t], 0, 0, 0);
                    }
#if KGWAS_COARSE_PHASES
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }

            // Per-row terms of the 16 rows whose accumulator registers this lane holds (row kg*4+jj of tile rt):
            // N1 from the ones column (slot 15 of the last column group: lane kg*16+15 holds it for these rows),
            // sqrt(d) rounded down (+inf for a row that does not exist or fails the MAC filter: nothing survives),
            // and the row's error term E(N1) rounded up.
            float sqd[RT * 4], er[RT * 4];
            {
                const uint64_t left = a.n_rows - rbase;
                const uint32_t rows_here = left < 64u ? (uint32_t)left : 64u;
                const bool mac_any = a.S >= 2u * a.min_count;  // else no N1 can satisfy mc <= N1 <= S - mc
                const uint32_t span = a.S - 2u * a.min_count;
                uint32_t n_ok = 0;
#pragma unroll
                for (int i = 0; i < RT * 4; i++) {
                    const int rt = i >> 2, jj = i & 3;
                    const uint32_t n1r = (uint32_t)__shfl(acc[rt][T - 1][jj], (int)(kg * 16u + 15u));
                    const bool ok = mac_any & ((uint32_t)(rt * 16 + jj) + kg * 4u < rows_here) & ((n1r - a.min_count) <= span);
                    n_ok += ok ? 1u : 0u;
                    const float f = (float)n1r;
                    const float sq = __builtin_amdgcn_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; 1 ulp sqrt; (1 - 2^-20)
                    sqd[i] = ok ? sq : __builtin_huge_valf();
                    er[i] = (a.eg_max + fminf(a.rall_max, f * a.rmax_max)) * 1.000001f;
                    if (KGWAS_COARSE_ABLATE & 16) {
                        sqd[i] = 1000.0f + i;
                        er[i] = 1.0f;
                    }
                }
                if (lg == 0 && m == 15u) tested_local += n_ok;  // the four m = 15 lanes cover the wave's 64 rows
            }
            // A lane holds 16 (row, column) pairs per column group; the four kg lanes of one m share the column.
            // All groups are tested first; then, if the wave has survivors at all, each column's list counter is bumped
            // ONCE for the 64 pairs the wave holds for it, all the (returning) atomics are issued back to back, and
            // only then are the rows written: one L2 round trip per wave pass instead of one per group (that was 55 %
            // of the kernel), and no atomic per survivor (which serialised the early chunks in the L2).
            uint32_t mb[PG];
            uint32_t any_bits = 0;
#pragma unroll
            for (int g = 0; g < PG; g++) {
                const float al = colc[g * 16 + m], iu = colc[SLOTS + g * 16 + m];
                uint32_t mbits = 0;
#pragma unroll
                for (int i = RT * 4 - 1; i >= 0; i--) {
                    int dc = acc[i >> 2][NS * g][i & 3];
                    if (NS == 2) dc = dc * 254 + acc[i >> 2][NS * g + NS - 1][i & 3];
                    const float lim = fmaf(al, sqd[i], -(iu * er[i]));
                    mbits = (mbits << 1) | ((fabsf((float)dc) >= lim) ? 1u : 0u);
                }
                if (KGWAS_COARSE_ABLATE & 1) mbits = (acc[0][NS * g][0] == 0x7fffffff) ? 1u : 0u;  // keeps the accumulators alive
                mb[g] = mbits;
                any_bits |= mbits;
            }
            if (KGWAS_COARSE_ABLATE & 32) any_bits = (any_bits == 0x12345678u) ? 1u : 0u;
            if ((KGWAS_COARSE_ABLATE & 128) && lane == 0 && any_bits) atomicAdd(&sctl[0], 0u);
            if (!(KGWAS_COARSE_ABLATE & 128) && __any(any_bits != 0u)) {  // wave-uniform
                uint32_t lane_cnt = 0;
#pragma unroll
                for (int g = 0; g < PG; g++) lane_cnt += __popc(mb[g]);
                uint32_t incl = lane_cnt;  // inclusive scan over the wave
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(incl, d);
                    if ((int)lane >= d) incl += t;
                }
                const uint32_t total = __shfl(incl, 63);
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&sctl[0], total);  // LDS
                wbase = __shfl(wbase, 0);
                const bool fits = wbase + total <= COARSE_SBUF;  // wave-uniform
                uint32_t gb = 0;
                if (!fits) {  // dense survivors (ramp chunks): this wave appends to the global list itself
                    if (lane == 0) {
                        atomicMin(&sctl[1], wbase);
                        gb = atomicAdd(a.key_count, total);
                    }
                    gb = __shfl(gb, 0);
                }
                uint32_t k = (fits ? wbase : gb) + (incl - lane_cnt);
#pragma unroll
                for (int g = 0; g < PG; g++) {
                    uint32_t mbits = (KGWAS_COARSE_ABLATE & 64) ? 0u : mb[g];
                    const uint32_t pk = (uint32_t)colp[g * 16 + m] << a.row_bits;  // column >= 0 wherever a bit is set
                    while (mbits) {
                        const uint32_t b = __ffs(mbits) - 1u;
                        mbits &= mbits - 1u;
                        const uint32_t key = pk | (uint32_t)(rbase + (b >> 2) * 16u + kg * 4u + (b & 3u));
                        if (fits)
                            sbuf[k] = key;
                        else if (k < a.key_cap)
                            a.keys[k] = key;
                        k++;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = sctl[0] < sctl[1] ? sctl[0] : sctl[1];
        sctl[3] = n;
        sctl[2] = n ? atomicAdd(a.key_count, n) : 0u;
    }
    __syncthreads();
    {
        const uint32_t n = sctl[3], gb = sctl[2];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
            if (gb + i < a.key_cap) a.keys[gb + i] = sbuf[i];
    }
    if (a.tested) {
        uint32_t v = tested_local;
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane == 15u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// Exact re-scoring of the survivors of one phenotype column (blockIdx.y), one lane per survivor.
// Same arithmetic as score_valu_kernel: the reference's select-and-add chains, then finish_pair.
__global__ void __launch_bounds__(256) rescore_kernel(ScoreArgs a, const uint32_t* keys, const uint32_t* surv_off,
                                                      const uint32_t* surv_cnt, uint32_t surv_cap, uint32_t row_mask) {
    const uint32_t p = blockIdx.y;
    const uint32_t n = surv_cnt[p];
    if (n > surv_cap) return;  // overflow: the host redoes this chunk
    if (blockIdx.x * 256u >= n) return;  // block-uniform
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool valid = i < n;
    const uint64_t r = keys[surv_off[p] + (valid ? i : 0u)] & row_mask;
    const uint32_t* rp = a.src.base + r * a.src.stride_dw + a.src.off_dw;
    const uint32_t L = 64u * a.W_m;
    const uint32_t nblk = a.W_m / 2u;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t n1 = 0;
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t w[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint2 v = make_uint2(0u, 0u);
            if (4u * b + 2u * h + 1u < a.src.avail_dw) v = *reinterpret_cast<const uint2*>(rp + 4u * b + 2u * h);
            w[2 * h] = v.x & a.dmask[4 * b + 2 * h];
            w[2 * h + 1] = v.y & a.dmask[4 * b + 2 * h + 1];
        }
        n1 += __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
        const float* yb = a.Yperm + (size_t)p * L + 128u * b;
#pragma unroll
        for (int s = 0; s < 32; s++)
#pragma unroll
            for (int l = 0; l < 4; l++) {
                const int mk = ((int)(w[l] << s)) >> 31;
                acc[l] = acc[l] + __int_as_float(mk & __float_as_int(yb[4 * s + l]));
            }
    }
    if (!valid) return;
    const float yf = ((acc[0] + acc[1]) + acc[2]) + acc[3];
    if (!a.so_score) {
        finish_pair(a, r, p, yf, n1, mac_pass(a, n1), a.sums[p], a.thr[p]);
        return;
    }
    // Ordered mode: the survivor list is sorted by row, entry i goes to position i (coalesced).
    double q, d, s, out = -__builtin_huge_val();
    score_terms(a, yf, n1, a.sums[p], q, d);
    if (mac_pass(a, n1) && candidate_score(a, p, q, d, a.thr[p], s)) out = s;
    const uint64_t o = (uint64_t)p * surv_cap + i;
    a.so_score[o] = out;
    a.so_kmer[o] = a.file_rows[r * a.file_stride_w];
    a.so_row[o] = (uint32_t)r;
}

// B operands of one LDS group + the group's per-column constants (3 x up to 128 words) + the block's survivor buffer
size_t coarse_lds_bytes(uint32_t n_kgroups, uint32_t T) { return (size_t)n_kgroups * 8u * T * 1024u + 1536u + 16u + 4u * COARSE_SBUF; }

template <int T, int NS>
static hipError_t launch_coarse_t(const CoarseArgs& a, uint32_t rows_per_block, uint32_t n_rowblocks, size_t lds,
                                  hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)coarse_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((coarse_kernel<T, NS>), dim3(n_rowblocks), dim3(512), lds, st, a, rows_per_block, n_rowblocks);
    return hipGetLastError();
}

hipError_t launch_coarse(const CoarseArgs& a, uint32_t T, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const size_t lds = coarse_lds_bytes(a.n_kgroups, T);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    const uint32_t threads = 512;
    if ((a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit byte offsets
    const uint32_t rpp = (threads >> 6) * 64u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
#ifdef KGWAS_COARSE_BENCH_ONLY  // experiments: only the two shapes of the 1024 x 101 bench (fast compiles)
    if (a.n_slices == 1 && T == 7) return launch_coarse_t<7, 1>(a, rows_per_block, n_rowblocks, lds, st);
    if (a.n_slices == 2 && T == 8) return launch_coarse_t<8, 2>(a, rows_per_block, n_rowblocks, lds, st);
    return hipErrorInvalidValue;
#else
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

hipError_t launch_rescore(const ScoreArgs& a, const uint32_t* keys, const uint32_t* surv_off, const uint32_t* surv_cnt,
                          uint32_t surv_cap, uint32_t row_bits, hipStream_t st) {
    if (a.n_pheno == 0 || surv_cap == 0) return hipSuccess;
    const uint32_t row_mask = row_bits >= 32 ? 0xFFFFFFFFu : ((1u << row_bits) - 1u);
    hipLaunchKernelGGL(rescore_kernel, dim3((surv_cap + 255u) / 256u, a.n_pheno), dim3(256), 0, st, a, keys, surv_off, surv_cnt,
                       surv_cap, row_mask);
    return hipGetLastError();
}

}  // namespace kgwas
