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
#include "score_common.h"

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
//   |Dc| >= alpha_p * sqrt(d(N1)) - e_p(N1),    alpha_p = sqrt(thr_p)/(N*u),  e_p(N1) = (Eg + min(Rall, N1*rmax))/u
// (score_coarse.hip header for Eg, Rall, rmax). The right-hand side is evaluated in float32 from constants the host
// rounded in the safe direction (alpha down by 2^-19 relative, sqrt(d) down by 2^-20, the error terms up, plus an
// absolute pad), which dominates the float32 rounding of the three operations: the computed limit never exceeds the
// true one, so no possible candidate is dropped. 8 cheap lane-ops per pair instead of 14 double-precision ones.
template <int T, int NS>
__global__ void __launch_bounds__(512) coarse_kernel(CoarseArgs a, uint32_t rows_per_block, uint32_t n_rowblocks) {
    extern __shared__ i32x4 blds[];  // [n_kgroups][8][T][64] x 16 bytes, then float colc[4][PG*16]
    constexpr int PG = T / NS;       // column groups (of 16) per LDS group
    constexpr int RT = 4;
    const uint32_t rb = blockIdx.x;
    if (rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t kg = lane >> 4, m = lane & 15u;
    const uint32_t n_kgroups = a.n_kgroups;
    const uint32_t group_vec = n_kgroups * 8u * T * 64u;  // i32x4 elements per LDS group
    float* colc = reinterpret_cast<float*>(blds + group_vec);
    const uint32_t rows_per_pass = (blockDim.x >> 6) * (RT * 16u);
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    uint32_t tested_local = 0;

    for (uint32_t lg = 0; lg < a.n_lgroups; lg++) {
        if (lg) __syncthreads();
        {
            const i32x4* src = reinterpret_cast<const i32x4*>(a.Bq) + (size_t)lg * group_vec;
            for (uint32_t i = threadIdx.x; i < group_vec; i += blockDim.x) blds[i] = src[i];
            if (threadIdx.x < PG * 16u) {
                const uint32_t p = lg * (PG * 16u) + threadIdx.x;
                float al = __builtin_huge_valf(), eg = 0.0f, ra = 0.0f, rm = 0.0f;  // padding column: nothing survives
                if (p < a.n_pheno) {
                    const CoarseCol cc = a.cols[p];
                    al = (float)(sqrt(a.thr[p]) * cc.kalpha);  // NaN threshold (frozen column) -> NaN -> nothing survives
                    eg = cc.eg;
                    ra = cc.rall;
                    rm = cc.rmax;
                }
                colc[threadIdx.x] = al;
                colc[PG * 16u + threadIdx.x] = eg;
                colc[2u * PG * 16u + threadIdx.x] = ra;
                colc[3u * PG * 16u + threadIdx.x] = rm;
            }
        }
        __syncthreads();

        for (uint32_t ps = 0; ps * rows_per_pass < rows_per_block; ps++) {
            const uint64_t rbase = blk_row0 + (uint64_t)ps * rows_per_pass + wave * (RT * 16u);
            if (rbase >= a.n_rows) break;  // wave-uniform
            // 32-bit dword offsets from the (uniform) base: launch_coarse guarantees n_rows * stride < 2^32
            uint32_t ro[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint64_t r = rbase + rt * 16u + m;
                if (r >= a.n_rows) r = a.n_rows - 1;
                ro[rt] = (uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw;
            }
            i32x4 acc[RT][T];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int t = 0; t < T; t++) acc[rt][t] = (i32x4){0, 0, 0, 0};
            uint32_t n1p[RT] = {0u, 0u, 0u, 0u};

            for (uint32_t g = 0; g < n_kgroups; g++) {
                // this lane's 16 bytes of each row: dwords 16g + 4kg .. +3 (zero beyond the row's data, masked)
                uint32_t piece[RT][4];
                const uint32_t d0 = 16u * g + 4u * kg;
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        uint2 v = make_uint2(0u, 0u);
                        if (d0 + 2u * h + 1u < a.src.avail_dw)
                            v = *reinterpret_cast<const uint2*>(a.src.base + (ro[rt] + d0 + 2u * h));
                        piece[rt][2 * h] = v.x;
                        piece[rt][2 * h + 1] = v.y;
                    }
                }
                if (!a.all_ones) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t mk = (d0 + q < 2u * a.W_m) ? a.dmask[d0 + q] : 0u;
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) piece[rt][q] &= mk;
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    n1p[rt] += __popc(piece[rt][0]) + __popc(piece[rt][1]) + __popc(piece[rt][2]) + __popc(piece[rt][3]);

                const i32x4* bg = blds + (size_t)g * 8u * T * 64u + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    i32x4 A[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) A[rt] = expand16((piece[rt][j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        const i32x4 B = bg[(j * T + t) * 64];
#pragma unroll
                        for (int rt = 0; rt < RT; rt++)
                            acc[rt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt], B, acc[rt][t], 0, 0, 0);
                    }
                }
            }

            // N1 of each row: the four kg lanes hold disjoint pieces.
            uint32_t n1[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                uint32_t v = n1p[rt];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                n1[rt] = v;
            }
            if (lg == 0 && kg == 0) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const uint64_t r = rbase + rt * 16u + m;
                    if (r < a.n_rows && a.S >= a.min_count && n1[rt] >= a.min_count && n1[rt] <= a.S - a.min_count)
                        tested_local++;
                }
            }
            // Per-row terms of the 16 rows whose accumulator registers this lane holds (row kg*4+jj of tile rt):
            // N1, sqrt(d) rounded down, and whether the row exists and passes the MAC filter.
            float n1f[RT * 4], sqd[RT * 4];
            uint32_t rowok = 0;
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const uint32_t trow = kg * 4u + jj;
                    const uint32_t n1r = __shfl(n1[rt], (int)trow);
                    const uint64_t r = rbase + rt * 16u + trow;
                    if ((r < a.n_rows) && (a.S >= a.min_count) && (n1r >= a.min_count) && (n1r <= a.S - a.min_count))
                        rowok |= 1u << (rt * 4 + jj);
                    const float f = (float)n1r;
                    n1f[rt * 4 + jj] = f;
                    sqd[rt * 4 + jj] = __builtin_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; (1 - 2^-20)
                }
            }
            // A lane holds 16 (row, column p) pairs per column group, and the four kg lanes of one m share the
            // column: survivors are counted over those 64 pairs first and the column's counter is bumped once
            // (early chunks have ~1e6 survivors on ~100 counters; one atomic per survivor serialises in the L2).
#pragma unroll
            for (int g = 0; g < PG; g++) {
                const uint32_t p = (lg * PG + g) * 16u + m;
                const float al = colc[g * 16 + m], eg = colc[PG * 16 + g * 16 + m], ra = colc[2 * PG * 16 + g * 16 + m],
                            rm = colc[3 * PG * 16 + g * 16 + m];
                uint32_t mbits = 0;
#pragma unroll
                for (int i = 0; i < RT * 4; i++) {
                    int dc = acc[i >> 2][NS * g][i & 3];
                    if (NS == 2) dc = dc * 254 + acc[i >> 2][NS * g + NS - 1][i & 3];
                    const float lim = fmaf(al, sqd[i], -eg) - fminf(ra, n1f[i] * rm);
                    if (fabsf((float)dc) >= lim) mbits |= 1u << i;
                }
                mbits &= rowok;
                if (__any(mbits != 0u)) {  // wave-uniform
                    const uint32_t cnt = __popc(mbits);
                    const uint32_t c0 = __shfl(cnt, (int)m), c1 = __shfl(cnt, (int)(m + 16u)), c2 = __shfl(cnt, (int)(m + 32u)),
                                   c3 = __shfl(cnt, (int)(m + 48u));
                    const uint32_t total = c0 + c1 + c2 + c3;
                    uint32_t base = 0;
                    if (kg == 0 && total) base = atomicAdd(&a.surv_cnt[p], total);
                    base = __shfl(base, (int)m);
                    uint32_t slot = base + (kg > 0 ? c0 : 0u) + (kg > 1 ? c1 : 0u) + (kg > 2 ? c2 : 0u);
                    while (mbits) {
                        const uint32_t b = __ffs(mbits) - 1u;
                        mbits &= mbits - 1u;
                        const uint64_t r = rbase + (b >> 2) * 16u + kg * 4u + (b & 3u);
                        if (slot < a.surv_cap) a.surv[(uint64_t)p * a.surv_cap + slot] = (uint32_t)r;
                        slot++;
                    }
                }
            }
        }
    }
    if (a.tested) {
        uint32_t v = tested_local;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (lane == 0 && v) atomicAdd(a.tested, (unsigned long long)v);
    }
}

// Exact re-scoring of the survivors of one phenotype column (blockIdx.y), one lane per survivor.
// Same arithmetic as score_valu_kernel: the reference's select-and-add chains, then finish_pair.
__global__ void __launch_bounds__(256) rescore_kernel(ScoreArgs a, const uint32_t* surv, const uint32_t* surv_cnt,
                                                      uint32_t surv_cap) {
    const uint32_t p = blockIdx.y;
    uint32_t n = surv_cnt[p];
    if (n > surv_cap) {
        if (a.so_score) return;  // overflow: the list was not sorted; the host redoes this chunk
        n = surv_cap;
    }
    if (blockIdx.x * 256u >= n) return;  // block-uniform
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool valid = i < n;
    const uint64_t r = surv[(uint64_t)p * surv_cap + (valid ? i : 0u)];
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

// B operands of one LDS group + the group's per-column float constants (4 x up to 128 floats)
size_t coarse_lds_bytes(uint32_t n_kgroups, uint32_t T) { return (size_t)n_kgroups * 8u * T * 1024u + 2048u; }

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
    if (a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit row offsets
    const uint32_t rpp = (threads >> 6) * 64u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
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
}

hipError_t launch_rescore(const ScoreArgs& a, const uint32_t* surv, const uint32_t* surv_cnt, uint32_t surv_cap,
                          hipStream_t st) {
    if (a.n_pheno == 0 || surv_cap == 0) return hipSuccess;
    hipLaunchKernelGGL(rescore_kernel, dim3((surv_cap + 255u) / 256u, a.n_pheno), dim3(256), 0, st, a, surv, surv_cnt, surv_cap);
    return hipGetLastError();
}

}  // namespace kgwas
