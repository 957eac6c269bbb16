// score_mfma.hip — exact-order scorer on the MI355X matrix cores (v_mfma_f32_16x16x4_f32).
//
// calculate_kmer_score (src/kmers_multiple_databases.cpp:327-363) accumulates, per (k-mer,
// phenotype), four independent float32 chains: SSE lane l walks samples 128b+32l+31-s for
// b = 0.., s = 0..31 (:344-356). Adding (bit ? y : +0.0f) is bit-identical to
// fmaf((float)bit, y, acc) for finite y, and gfx950's f32-input MFMA is bit-for-bit a k-ordered
// fmaf chain, so the reference's exact scores come straight out of the matrix pipe:
//
//   block  = 8 waves sharing NCT (1 or 2) 16-column phenotype tiles in LDS, each [L][16] floats in
//            chain-step order, four consecutive t per lane contiguous (one ds_read_b128):
//            (((b*4+l)*2 + t/4)*64 + kk*16+n)*4 + t%4  ->  y_n[128b+32l+31-(4t+kk)]
//   wave   = 2 row tiles of 16 k-mers x NCT column tiles per pass -> 8*NCT independent accumulators
//   lane   = (m = lane & 15: k-mer row of the tile, kk = lane >> 4: k index 0..3)
//   A[m][kk] = bit 31-(4t+kk) of the row's 32-bit SSE-lane word, as 0.0f / 1.0f
//   B[kk][n] = y_n[128b+32l+31-(4t+kk)]  (k index of the MFMA = kk)
//   D reg j  = row (lane>>4)*4 + j, column lane & 15
//
// What the probes (tools/probe_score.hip) showed and this layout answers:
//   * The f32-input MFMA runs at the vector-ALU rate and shares the SIMD with every other
//     instruction the waves issue: time per MFMA ~ 32 cycles + ~5 cycles per non-MFMA instruction,
//     whatever the occupancy. So the design goal is instructions per MFMA, not latency hiding:
//     - every A operand is used for NCT column tiles (N-blocking) and every B operand for both row
//       tiles: 2 A + NCT B per 2*NCT MFMAs;
//     - A operands are built 4 at a time: (w >> (7-kk)) & 0x01010101 puts the bits of t = 0,2,4,6
//       in the four bytes of one register, (w >> (3-kk)) those of t = 1,3,5,7, and
//       v_cvt_f32_ubyte{3..0} turns each byte into 0.0f/1.0f: 12 VALU ops per 8 operands, built for
//       a whole 128-sample block before its MFMA stream (no VALU->MFMA wait states in the stream).
//   * Table rows are read in place, in their on-disk layout. The four kk-lanes of a row fetch four
//     DIFFERENT 16-byte pieces (64 contiguous bytes of the row per instruction, every touched cache
//     line used in full, once) and the words are broadcast to the row's four lanes with two VALU lane
//     swaps (v_permlane32_swap + v_permlane16_swap); the next 64 bytes are in flight meanwhile.
//   * One block = one row block (rows_per_block rows, ~140 KB of table) walked once per column-tile
//     group with the group's phenotype tiles swapped into LDS in between: all blocks do equal work,
//     and the table is read from HBM once and re-read from that XCD's L2 for the later groups.
#include "score_common.h"

namespace kgwas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef KGWAS_MFMA_MAX_THREADS
#define KGWAS_MFMA_MAX_THREADS 512
#endif

// x holds, in the lanes of 16-lane row kk, the value X_kk(m). Returns X_0..X_3 in every lane.
//   permlane32_swap(x, x)   -> { [X0 X1 X0 X1], [X2 X3 X2 X3] }   (rows 0..3 of the wave)
//   permlane16_swap(y, y)   -> { [Y0 Y0 Y2 Y2], [Y1 Y1 Y3 Y3] }
__device__ __forceinline__ void bcast_rows(uint32_t x, uint32_t (&out)[4]) {
    const u32x2 s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    const u32x2 a = __builtin_amdgcn_permlane16_swap(s[0], s[0], false, false);
    const u32x2 b = __builtin_amdgcn_permlane16_swap(s[1], s[1], false, false);
    out[0] = a[0];
    out[1] = a[1];
    out[2] = b[0];
    out[3] = b[1];
}

__device__ __forceinline__ uint4 load_piece(const uint32_t* p) {  // 8-byte aligned, 16 bytes
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 2);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// One 128-sample block of the chain for both row tiles and NCT column tiles: 64*NCT MFMAs.
// w[rt][l]: the row's SSE-lane word l (un-shifted); sh_e = 7-kk, sh_o = 3-kk.
// yb: this block's B values for this lane in column tile 0; tile c is ct_stride floats further.
template <int NCT>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[2][NCT][4], const uint32_t (&w)[2][4], uint32_t sh_e,
                                           uint32_t sh_o, const float* yb, uint32_t ct_stride) {
    float Af[2][4][8];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int l = 0; l < 4; l++) {
            uint32_t e = (w[rt][l] >> sh_e) & 0x01010101u;  // bytes 3..0: t = 0,2,4,6
            uint32_t o = (w[rt][l] >> sh_o) & 0x01010101u;  // bytes 3..0: t = 1,3,5,7
            // Keep e/o as materialised registers: otherwise the compiler folds the byte selects back
            // into one v_bfe_u32 + v_cvt_f32_ubyte0 per operand instead of a lone v_cvt_f32_ubyteN.
            asm volatile("" : "+v"(e), "+v"(o));
#pragma unroll
            for (int h = 0; h < 4; h++) {
#if defined(KGWAS_ABLATE) && (KGWAS_ABLATE & 16)  // probe: operands without the per-operand conversion
                Af[rt][l][2 * h] = __int_as_float(e);
                Af[rt][l][2 * h + 1] = __int_as_float(o);
#else
                Af[rt][l][2 * h] = (float)((e >> (8 * (3 - h))) & 0xFFu);      // v_cvt_f32_ubyteN
                Af[rt][l][2 * h + 1] = (float)((o >> (8 * (3 - h))) & 0xFFu);
#endif
            }
        }
    __builtin_amdgcn_sched_barrier(0);  // operands first, then a clean MFMA stream
    // B operands: one ds_read_b128 per (SSE lane l, half th, tile c) brings this lane's values for
    // four consecutive t; each chain (rt, c, l) still sees t = 0..7 in ascending order.
#pragma unroll
    for (int th = 0; th < 2; th++)
#pragma unroll
        for (int l = 0; l < 4; l++) {
            f32x4 Bq[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++)
                Bq[c] = *reinterpret_cast<const f32x4*>(yb + c * ct_stride + (l * 2 + th) * 256);
#pragma unroll
            for (int tq = 0; tq < 4; tq++)
#pragma unroll
                for (int c = 0; c < NCT; c++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][c][l] = __builtin_amdgcn_mfma_f32_16x16x4f32(Af[rt][l][4 * th + tq], Bq[c][tq],
                                                                            acc[rt][c][l], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NCT>
__device__ __forceinline__ void score_rows(const ScoreArgs& a, const float* ylds, uint32_t ct0, uint64_t blk_row0,
                                           uint32_t rows_per_block, uint32_t nb_full) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t kk = lane >> 4;
    const uint32_t m = lane & 15u;
    const uint32_t sh_e = 7u - kk, sh_o = 3u - kk;
    const uint32_t L = 64u * a.W_m;
    const uint32_t ct_stride = L * 16u;
    const uint32_t nblk = a.W_m / 2u;
    const uint32_t ng_full = nb_full / 4u;  // groups of four unmasked, fully present blocks
    const uint32_t rows_per_pass = (blockDim.x >> 6) * 32u;

    // This lane's phenotype columns are fixed for the whole block: keep their constants in registers.
    uint32_t p[NCT];
    bool has_p[NCT];
    float sum_p[NCT];
    double thr_p[NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) {
        p[c] = (ct0 + c) * 16u + m;
        has_p[c] = p[c] < a.n_pheno;
        sum_p[c] = has_p[c] ? a.sums[p[c]] : 0.0f;
        thr_p[c] = (has_p[c] && a.thr) ? a.thr[p[c]] : 0.0;
    }
    uint32_t tested_local = 0;

    for (uint32_t ps = 0; ps * rows_per_pass < rows_per_block; ps++) {
        const uint64_t rbase = blk_row0 + (uint64_t)ps * rows_per_pass + wave * 32u;
        if (rbase >= a.n_rows) break;  // wave-uniform

        const uint32_t* rp[2];
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            uint64_t r = rbase + rt * 16u + m;
            if (r >= a.n_rows) r = a.n_rows - 1;  // clamp loads; results discarded below
            rp[rt] = a.src.base + r * a.src.stride_dw + a.src.off_dw;
        }

        f32x4 acc[2][NCT][4];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int c = 0; c < NCT; c++)
#pragma unroll
                for (int l = 0; l < 4; l++) acc[rt][c][l] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t n1p[2] = {0u, 0u};  // popcount of this lane's own pieces (disjoint across the 4 kk lanes)
        uint32_t n1t[2] = {0u, 0u};  // popcount of tail blocks (identical in the 4 kk lanes)

        // ---- full groups: 4 blocks = 64 bytes per row; lane kk holds block 4g+kk ---------------
        uint4 nx[2];
        if (ng_full) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++) nx[rt] = load_piece(rp[rt] + 4u * kk);
        }
        for (uint32_t g = 0; g < ng_full; g++) {
            uint32_t cur[2][4];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                cur[rt][0] = nx[rt].x;
                cur[rt][1] = nx[rt].y;
                cur[rt][2] = nx[rt].z;
                cur[rt][3] = nx[rt].w;
            }
            if (g + 1 < ng_full) {  // prefetch the next group while this one's MFMAs issue
#pragma unroll
                for (int rt = 0; rt < 2; rt++) nx[rt] = load_piece(rp[rt] + 16u * (g + 1) + 4u * kk);
            }
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
                n1p[rt] += __popc(cur[rt][0]) + __popc(cur[rt][1]) + __popc(cur[rt][2]) + __popc(cur[rt][3]);
            // all[rt][l][qb] = SSE-lane word l of block 4g+qb (held by lane row qb), in every lane
            uint32_t all[2][4][4];
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int l = 0; l < 4; l++) bcast_rows(cur[rt][l], all[rt][l]);
#pragma unroll
            for (int qb = 0; qb < 4; qb++) {
                uint32_t w[2][4];
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int l = 0; l < 4; l++) w[rt][l] = all[rt][l][qb];
                mfma_block<NCT>(acc, w, sh_e, sh_o, ylds + (size_t)(4u * g + qb) * 2048u + lane * 4u, ct_stride);
            }
        }
        // ---- tail blocks: partial masks and/or dwords beyond the row's data --------------------
        for (uint32_t b = 4u * ng_full; b < nblk; b++) {
            uint32_t w[2][4];
            const bool have_lo = (4u * b + 1u) < a.src.avail_dw;  // avail_dw is even
            const bool have_hi = (4u * b + 3u) < a.src.avail_dw;
            const uint32_t k0 = a.dmask[4 * b + 0], k1 = a.dmask[4 * b + 1];
            const uint32_t k2 = a.dmask[4 * b + 2], k3 = a.dmask[4 * b + 3];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
                if (have_lo) lo = *reinterpret_cast<const uint2*>(rp[rt] + 4u * b);
                if (have_hi) hi = *reinterpret_cast<const uint2*>(rp[rt] + 4u * b + 2u);
                w[rt][0] = lo.x & k0;
                w[rt][1] = lo.y & k1;
                w[rt][2] = hi.x & k2;
                w[rt][3] = hi.y & k3;
                n1t[rt] += __popc(w[rt][0]) + __popc(w[rt][1]) + __popc(w[rt][2]) + __popc(w[rt][3]);
            }
            mfma_block<NCT>(acc, w, sh_e, sh_o, ylds + (size_t)b * 2048u + lane * 4u, ct_stride);
        }

        // N1 of each row: the four kk lanes' disjoint pieces plus the common tail.
        uint32_t n1[2];
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            uint32_t v = n1p[rt];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            n1[rt] = v + n1t[rt];
        }

        // Row bookkeeping comes from the kk == 0 lanes (lane == row of the tile).
        if (ct0 == 0 && kk == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const uint64_t r = rbase + rt * 16u + m;
                if (r < a.n_rows) {
                    if (a.n1_out) a.n1_out[r] = n1[rt];
                    if (a.kmer_out) a.kmer_out[r] = a.file_rows[r * a.file_stride_w];
                    tested_local += mac_pass(a, n1[rt]) ? 1u : 0u;
                }
            }
        }

#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t trow = kk * 4u + j;  // D row held in register j
                const uint32_t n1r = __shfl(n1[rt], (int)trow);
                const uint64_t r = rbase + rt * 16u + trow;
                const bool pass = mac_pass(a, n1r);
#pragma unroll
                for (int c = 0; c < NCT; c++) {
                    const float yf =
                        ((acc[rt][c][0][j] + acc[rt][c][1][j]) + acc[rt][c][2][j]) + acc[rt][c][3][j];  // :358
#if defined(KGWAS_ABLATE) && (KGWAS_ABLATE & 1)
                    asm volatile("" ::"v"(yf), "v"(n1r), "v"(r));
#else
                    if (r < a.n_rows && has_p[c]) finish_pair(a, r, p[c], yf, n1r, pass, sum_p[c], thr_p[c]);
#endif
                }
            }
        }
    }

    // One atomic per wave for the tested-k-mers count (all lanes are active here).
    if (ct0 == 0 && a.tested) {
        uint32_t v = tested_local;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (lane == 0 && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// NCTMAX = column tiles per block (2 when two tiles fit the CU's 160 KB of LDS, else 1).
template <int NCTMAX>
__global__ void __launch_bounds__(KGWAS_MFMA_MAX_THREADS)
    score_mfma_kernel(ScoreArgs a, uint32_t rows_per_block, uint32_t n_rowblocks, uint32_t n_ctiles, uint32_t nb_full) {
    extern __shared__ float ylds[];  // [NCT][L][16]
    // Every block owns one row block and walks ALL column-tile groups for it, so all blocks do the
    // same amount of work (blocks of a lone odd tile would otherwise cost as much as paired ones:
    // workgroups land on CUs round-robin, and a group index periodic with the 32 CUs of an XCD
    // pins the short blocks to a fixed subset of CUs). The row block (rows_per_block * row bytes,
    // ~140 KB) is re-read from L2 for each group.
    const uint32_t rb = blockIdx.x;
    if (rb >= n_rowblocks) return;
    const uint32_t n_groups = (n_ctiles + NCTMAX - 1) / NCTMAX;
    const uint32_t L = 64u * a.W_m;
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    for (uint32_t grp = 0; grp < n_groups; grp++) {
        const uint32_t ct0 = grp * NCTMAX;
        const uint32_t nct_here = (n_ctiles - ct0 < (uint32_t)NCTMAX) ? (n_ctiles - ct0) : (uint32_t)NCTMAX;
        if (grp) __syncthreads();  // everyone is done reading the previous group's tiles
        {
            const float4* src = reinterpret_cast<const float4*>(a.Ymfma + (size_t)ct0 * L * 16u);
            float4* dst = reinterpret_cast<float4*>(ylds);
            for (uint32_t i = threadIdx.x; i < nct_here * L * 4u; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
        if (NCTMAX == 2 && nct_here == 2)
            score_rows<NCTMAX>(a, ylds, ct0, blk_row0, rows_per_block, nb_full);
        else
            score_rows<1>(a, ylds, ct0, blk_row0, rows_per_block, nb_full);
    }
}

size_t mfma_lds_bytes(uint32_t W_m) { return (size_t)64u * W_m * 16u * sizeof(float); }

hipError_t launch_score_mfma(const ScoreArgs& a, uint32_t rows_per_block, uint32_t nb_full, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const size_t tile = mfma_lds_bytes(a.W_m);
    if (tile > 160u * 1024u) return hipErrorInvalidValue;
    const uint32_t n_ctiles = (a.n_pheno + 15u) / 16u;
#ifdef KGWAS_PROBE_NCT
    const uint32_t nctmax = KGWAS_PROBE_NCT;
#else
    const uint32_t nctmax = (n_ctiles >= 2 && 2 * tile <= 160u * 1024u) ? 2u : 1u;
#endif
    const size_t lds = tile * nctmax;
    // Up to 80 KB two 4-wave blocks share a CU; beyond that one 8-wave block keeps two waves per SIMD.
    const uint32_t threads = (lds > 80u * 1024u) ? 512u : 256u;
    const uint32_t rpp = (threads >> 6) * 32u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
    const uint32_t grid = n_rowblocks;
    hipError_t e;
    if (nctmax == 2) {
        if (lds > 64 * 1024 && (e = hipFuncSetAttribute((const void*)score_mfma_kernel<2>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(score_mfma_kernel<2>, dim3(grid), dim3(threads), lds, st, a, rows_per_block, n_rowblocks,
                           n_ctiles, nb_full);
    } else {
        if (lds > 64 * 1024 && (e = hipFuncSetAttribute((const void*)score_mfma_kernel<1>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(score_mfma_kernel<1>, dim3(grid), dim3(threads), lds, st, a, rows_per_block, n_rowblocks,
                           n_ctiles, nb_full);
    }
    return hipGetLastError();
}

}  // namespace kgwas
