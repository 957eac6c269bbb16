// score_narrow.hip — the sparse-phase filter for scans with one to four phenotype columns: the HBM-bound
// configuration of calculate_kmer_score (src/kmers_multiple_databases.cpp:327-363; 8*(1+W_f) bytes per row against
// 2*S flop per column, BASELINE.md row 2').
//
// With so few columns the int8 filter of score_coarse.hip is bound by its operand expansion (two lane-ops per four
// table bits) and by the exchange its test needs. Here
//  * the table bits become an FP4 (E2M1) operand of gfx950's block-scaled MFMA: an FP4 nibble with only bit 0, 1 or 2
//    set is 0.5, 1.0 or 2.0, so `dword & (0x11111111 << j)` IS a valid operand of 8 values for j = 0, 1, 2 (one
//    lane-op per 8 table bits; the block scale of that operand, 2^0 / 2^-1 / 2^-2, exact, brings them all to 0.5) and
//    only bit 3, the FP4 sign bit, has to be shifted first: five lane-ops per 32 table bits instead of eight;
//  * the 16 "columns" of the one MFMA tile a row needs anyway carry, per phenotype column p < 4, THREE FP8 (E4M3)
//    slices of small integers,   y_i - c ~ sum_k u_k q_ki,  q_ki in [-15, 15],  u_{k+1} = u_k / 30,  c = sum / N
//    (residual below 4e-5 of the column's range), and a ones row (its dot product is the masked popcount N1):
//    operand row 4p + k = slice k of column p, row 4p + 3 = ones;
//  * the phenotype slices are the A operand and the table rows the B operand, so that the accumulator tile is
//    [slice rows] x [table rows]: lane (r = lane & 15, p = lane >> 4) ends up with exactly the four numbers its pair
//    (table row r, phenotype column p) needs - D_0, D_1, D_2 and N1 - and tests it on the spot: no exchange through
//    LDS, no second pass over the accumulators.
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies exact small values and accumulates in float32: every partial sum is a
// multiple of 0.5 below 2^23, i.e. exact, whatever the order. The test, per pair: with
//        r_c = N * sum_k u_k D_k + N1 * (N c - sum),     d = N1 (N - N1),
// keep the pair iff (|r_c| + N * E(N1) + pad)^2 >= thr * d * (1 - 2^-30), E(N1) = Eg + min(Rall, N1 * rmax) bounding
// |yigi_ref - yc| exactly as in score_coarse.hip (float32 summation error of the reference's chains + quantisation
// residuals); a float32 pre-screen with its own slack goes first, the double-precision form only runs for the waves
// that have a pair near its threshold. Survivors go to the same key list as the coarse filter's and are re-scored
// exactly (rescore_kernel), so results are bit-identical to the exact scorers'.
//
// Operand maps (measured on the device, tools/probe_mx.hip): the FP4 operand of lane (i = lane & 15, kb = lane >> 4)
// holds k = 32 kb + e in nibble e; the FP8 operand of lane (i, kb) holds k = 16 kb + e in byte e < 16 and
// k = 64 + 16 kb + (e - 16) in byte e >= 16; C/D: column lane & 15, rows 4 (lane >> 4) + i. A lane loads ITS OWN 16
// bytes of a table row (samples 512 g + 128 kb + 0..127 of sample group g) and MFMA step j = 0..3 takes bit 4 e' + j
// of dword q as k = 32 kb + 8 q + e': no data moves between lanes.
#include <stdlib.h>

#include <algorithm>

#include "score_common.h"

// Timing experiments only (wrong results; tools/variants.sh builds the variants). Bits: 1 no MFMAs (an XOR keeps the
// operands alive), 2 no operand expansion, 4 no test, 8 no piece reads from LDS, 16 no slice operand reads, 32 no global
// row loads.
#ifndef KGWAS_NARROW_ABLATE
#define KGWAS_NARROW_ABLATE 0
#endif

namespace kgwas {

typedef int nv8i __attribute__((ext_vector_type(8)));
typedef float nv4f __attribute__((ext_vector_type(4)));

namespace {
constexpr int NRT = 4;  // tiles of 16 table rows per wave pass
}  // namespace

// One 512-sample group of a wave pass: four MFMA steps over the lanes' 16-byte pieces of the table rows.
__device__ __forceinline__ void narrow_group(const uint32_t (&piece)[NRT][4], const uint4* sg, nv4f (&acc)[NRT]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint4 s0, s1;
        if (KGWAS_NARROW_ABLATE & 16) {
            s0 = make_uint4(j, 1, 2, 3);
            s1 = make_uint4(4, 5, 6, j);
        } else {
            s0 = sg[j * 128];
            s1 = sg[j * 128 + 1];
        }
        const nv8i Y = {(int)s0.x, (int)s0.y, (int)s0.z, (int)s0.w, (int)s1.x, (int)s1.y, (int)s1.z, (int)s1.w};
#pragma unroll
        for (int rt = 0; rt < NRT; rt++) {
            nv8i G = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (KGWAS_NARROW_ABLATE & 2)
                    G[q] = (int)piece[rt][q];
                else
                    G[q] = j < 3 ? (int)(piece[rt][q] & (0x11111111u << j)) : (int)((piece[rt][q] >> 1) & 0x44444444u);
            }
            if (KGWAS_NARROW_ABLATE & 1) {
                acc[rt][j] += __int_as_float((G[0] ^ G[1] ^ G[2] ^ G[3] ^ Y[j] ^ Y[7 - j]) & 0x3fffff);
                continue;
            }
            // A: the slices, FP8 E4M3 (cbsz 0), block scale 2^0; B: the table bits, FP4 (blgp 4), block scale
            // 2^0 / 2^-1 / 2^-2 / 2^-2 for nibble bit 0 / 1 / 2 / 2 (bit 3 shifted down by one)
            if (j == 0) acc[rt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Y, G, acc[rt], 0, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (j == 1) acc[rt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Y, G, acc[rt], 0, 4, 0, 0x7F7F7F7F, 0, 0x7E7E7E7E);
            if (j >= 2) acc[rt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Y, G, acc[rt], 0, 4, 0, 0x7F7F7F7F, 0, 0x7D7D7D7D);
        }
    }
}

// A lane's per-column constants (column p = lane >> 4), read once per block.
struct NarrowLane {
    float wf0, wf1, wf2, slackf, thrf;
    double thrd;  // the column's threshold as the launch found it (read once per block: the exact test must not wait for a load)
    bool live;    // p < n_pheno
    // survivors per (column, 65 536-row segment) are added up per BLOCK in LDS (a block's rows touch at most two segments) and
    // reach the chunk's counters with one atomic per block, column and segment: in the first chunks of a scan - thresholds still
    // low, a survivor in nearly every wave pass - every pass's own atomic landed on the few words of a one-column chunk
    uint32_t* lcnt;  // [NARROW_MAX_COLS][2] in LDS
    uint32_t seg0;   // the block's first segment
};

// Lane (r = lane & 15, p = lane >> 4) holds D_0, D_1, D_2 (units of 0.5) and N1 / 2 of pair (table row rt * 16 + r,
// column p) in acc[rt]: MAC predicate, float32 pre-screen, the exact double-precision test where a wave has a pair near
// its threshold, survivor keys.
__device__ __forceinline__ void narrow_test(const NarrowArgs& a, const NarrowCol* cols, const NarrowLane& L, const nv4f (&acc)[NRT],
                                            uint32_t lane, uint64_t rbase, bool mac_any, uint32_t span, uint32_t& tested_local) {
    const uint32_t r = lane & 15u, p = lane >> 4;
    const uint64_t left = a.n_rows - rbase;
    const uint32_t rows_here = left < 64u ? (uint32_t)left : 64u;
    const float Nf = (float)a.S;
    if (a.pack1) {
        // ONE column, its slices in all four column slots of the operand: the four lane groups hold the same numbers for
        // every row tile, so lane (r, kb) takes row tile kb's - the wave tests its 64 rows in one go instead of four
        // (a quarter of the pre-screen's instructions: 1.2 of the 26 ms a pass over 1.2 G rows takes), and a ballot IS
        // the pass's bitmap word (bit = lane = row).
        const uint32_t kb = p;
        const float d0 = kb == 0u ? acc[0][0] : kb == 1u ? acc[1][0] : kb == 2u ? acc[2][0] : acc[3][0];
        const float d1 = kb == 0u ? acc[0][1] : kb == 1u ? acc[1][1] : kb == 2u ? acc[2][1] : acc[3][1];
        const float d2 = kb == 0u ? acc[0][2] : kb == 1u ? acc[1][2] : kb == 2u ? acc[2][2] : acc[3][2];
        const float on = kb == 0u ? acc[0][3] : kb == 1u ? acc[1][3] : kb == 2u ? acc[2][3] : acc[3][3];
        const uint32_t n1 = (uint32_t)(2.0f * on);
        const bool ok = mac_any && (lane < rows_here) && ((n1 - a.min_count) <= span);
        tested_local += ok ? 1u : 0u;
        const float ycf = L.wf0 * d0 + L.wf1 * d1 + L.wf2 * d2;
        const float lf = (fabsf(Nf * ycf) + L.slackf) * 1.001f;
        const float f1 = (float)n1;
        const bool maybe1 = ok && !(lf * lf < L.thrf * (f1 * (Nf - f1)) * 0.998f);
        if (__ballot(maybe1)) {
            bool hit = false;
            if (maybe1) {
                const NarrowCol& cc = cols[0];
                const double Nd = (double)a.S, thr = L.thrd;
                const double N1 = (double)(2.0f * on);
                const double yc = cc.w[0] * (double)d0 + cc.w[1] * (double)d1 + cc.w[2] * (double)d2;
                const double rc = Nd * yc + N1 * cc.t1;
                const double e = cc.eg + fmin(cc.rall, N1 * cc.rmax);
                const double lhs = (fabs(rc) + Nd * e) * (1.0 + 0x1p-30) + cc.pad;
                hit = lhs * lhs >= thr * (N1 * (Nd - N1)) * (1.0 - 0x1p-30);  // NaN threshold: never
            }
            const unsigned long long w = __ballot(hit);
            if (w && lane == 0u) {
                const uint64_t word = (a.row_off + rbase) >> 6;
                a.bitmap[word] = w;
                if (a.seg_cnt) atomicAdd(&L.lcnt[(uint32_t)(word >> 10) - L.seg0], (uint32_t)__popcll(w));
            }
        }
        return;
    }
    uint32_t maybe = 0;  // bit rt: this lane's pair of row tile rt may pass
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
        const uint32_t n1 = (uint32_t)(2.0f * acc[rt][3]);  // the ones row, in units of 0.5
        const bool ok = mac_any && (rt * 16u + r < rows_here) && ((n1 - a.min_count) <= span);
        tested_local += (ok && p == 0u) ? 1u : 0u;
        // float32 pre-screen: |r| <= N |yc| (1 + 2^-10) + slack (every rounding and N1 |N c - sum|, N E, pad in slackf)
        const float ycf = L.wf0 * acc[rt][0] + L.wf1 * acc[rt][1] + L.wf2 * acc[rt][2];
        const float lf = (fabsf(Nf * ycf) + L.slackf) * 1.001f;
        const float f1 = (float)n1;
        if (ok && L.live && !(lf * lf < L.thrf * (f1 * (Nf - f1)) * 0.998f)) maybe |= 1u << rt;  // NaN threshold: decided below
    }
    if (__ballot(maybe != 0u)) {  // rare: about one wave pass in a hundred at one column
        uint32_t hit = 0;
        if (maybe) {
            const NarrowCol& cc = cols[p];
            const double Nd = (double)a.S, thr = L.thrd;
#pragma unroll
            for (int rt = 0; rt < NRT; rt++)
                if (maybe & (1u << rt)) {
                    const double N1 = (double)(2.0f * acc[rt][3]);
                    const double yc = cc.w[0] * (double)acc[rt][0] + cc.w[1] * (double)acc[rt][1] + cc.w[2] * (double)acc[rt][2];
                    const double rc = Nd * yc + N1 * cc.t1;
                    const double e = cc.eg + fmin(cc.rall, N1 * cc.rmax);
                    const double lhs = (fabs(rc) + Nd * e) * (1.0 + 0x1p-30) + cc.pad;
                    if (lhs * lhs >= thr * (N1 * (Nd - N1)) * (1.0 - 0x1p-30)) hit |= 1u << rt;  // NaN threshold: never
                }
        }
        // Survivors as a bitmap: one 64-bit word per (column, wave pass) - bit 16 rt + r = table row rt * 16 + r of the
        // pass. A ballot per row tile holds the four columns' 16 rows side by side; lane 0 puts the words together and
        // stores the non-zero ones (the chunk's bitmap is zeroed beforehand). Row order is then a property of the
        // bitmap, and the ordered key lists come out of a popcount scan (bitmap_*_kernel) instead of a radix sort.
        unsigned long long bal[NRT];
        unsigned long long any = 0;
#pragma unroll
        for (int rt = 0; rt < NRT; rt++) {
            bal[rt] = __ballot((hit >> rt) & 1u);
            any |= bal[rt];
        }
        if (any && lane == 0u) {
            const uint64_t word = (a.row_off + rbase) >> 6;
            for (uint32_t q = 0; q < a.n_pheno; q++) {
                unsigned long long w = 0;
#pragma unroll
                for (int rt = 0; rt < NRT; rt++) w |= ((bal[rt] >> (16u * q)) & 0xFFFFull) << (16 * rt);
                if (w) {
                    a.bitmap[(uint64_t)q * a.words_per_col + word] = w;
                    // survivors per (column, 65 536-row segment): what launch_narrow_keys' blocks add up instead of a count kernel
                    if (a.seg_cnt) atomicAdd(&L.lcnt[q * 2u + (uint32_t)(word >> 10) - L.seg0], (uint32_t)__popcll(w));
                }
            }
        }
    }
}

// the block's segment counters leave for the chunk's (the caller has a barrier between the block's last pass and this)
__device__ __forceinline__ void narrow_flush_counts(const NarrowArgs& a, const NarrowLane& L) {
    if (a.seg_cnt && threadIdx.x < 2u * NARROW_MAX_COLS) {
        const uint32_t c = L.lcnt[threadIdx.x], q = threadIdx.x >> 1, sg = L.seg0 + (threadIdx.x & 1u);
        if (c && q < a.n_pheno && sg < a.n_segs) atomicAdd(&a.seg_cnt[q * a.n_segs + sg], c);
    }
}

__device__ __forceinline__ NarrowLane narrow_lane_constants(const NarrowArgs& a, const NarrowCol* cols, uint32_t lane, uint32_t* lcnt, uint64_t blk_row0) {
    NarrowLane L;
    L.lcnt = lcnt;
    L.seg0 = (uint32_t)((a.row_off + blk_row0) >> 16);
    const uint32_t p = a.pack1 ? 0u : lane >> 4;  // (pack1: every lane group tests column 0, narrow_test)
    L.live = p < a.n_pheno;
    const NarrowCol& cc = cols[L.live ? p : 0u];
    L.wf0 = cc.wf[0];
    L.wf1 = cc.wf[1];
    L.wf2 = cc.wf[2];
    L.slackf = cc.slackf;
    L.thrd = a.thr[L.live ? p : 0u];
    L.thrf = (float)L.thrd;
    return L;
}

// Rows fetched by the lanes that use them: lane (r, kb) loads its own 16 bytes of table row r. Adjacent lanes then read
// addresses a row apart (136 bytes at 1024 samples): 64 separate requests per load instruction. The fall-back for
// launches whose rows are not 16-byte aligned or do not fit the staged kernel's LDS, and for a launch's last rows.
__global__ void __launch_bounds__(256) narrow_kernel(NarrowArgs a, uint32_t rows_per_block) {
    extern __shared__ uint4 nlds[];  // [n_steps][64 lanes][2] slice operands, then the column constants
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // an SGPR: pass addresses are scalar
    const uint32_t kb = lane >> 4, m = lane & 15u;
    const uint32_t n_steps = a.n_kgroups * 4u;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.Bn);
        for (uint32_t i = threadIdx.x; i < n_steps * 128u; i += blockDim.x) nlds[i] = src[i];
    }
    NarrowCol* lcols = reinterpret_cast<NarrowCol*>(nlds + n_steps * 128u);
    if (threadIdx.x < a.n_pheno * (sizeof(NarrowCol) / 8u))
        reinterpret_cast<double*>(lcols)[threadIdx.x] = reinterpret_cast<const double*>(a.cols)[threadIdx.x];
    __shared__ uint32_t lcnt_s[2 * NARROW_MAX_COLS];
    if (threadIdx.x < 2u * NARROW_MAX_COLS) lcnt_s[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t blk_row0 = (uint64_t)blockIdx.x * rows_per_block;
    const NarrowLane L = narrow_lane_constants(a, lcols, lane, lcnt_s, blk_row0);
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    uint32_t tested_local = 0;
    const bool mac_any = a.S >= 2u * a.min_count;
    const uint32_t span = a.S - 2u * a.min_count;

    for (uint32_t ps = 0; (ps * 4u + wave) * 64u < rows_per_block; ps++) {
        const uint64_t rbase = blk_row0 + (uint64_t)(ps * 4u + wave) * 64u;
        if (rbase >= a.n_rows) break;  // wave-uniform
        uint64_t ro[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; rt++) {
            uint64_t r = rbase + rt * 16u + m;
            if (r >= a.n_rows) r = a.n_rows - 1;
            ro[rt] = (r * a.src.stride_dw + a.src.off_dw) * 4ull;
        }
        nv4f acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; rt++) acc[rt] = (nv4f){0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t piece[NRT][4];
        for (uint32_t g = 0; g < a.n_kgroups; g++) {
            // beyond the row's data the loads are clamped onto its last 8 bytes: whatever bits arrive there meet zero
            // operands (sample slots >= S are zero in every slice)
            uint32_t b0 = 64u * g + 16u * kb, b1 = b0 + 8u;
            b0 = b0 + 8u <= avail_b ? b0 : avail_b - 8u;
            b1 = b1 + 8u <= avail_b ? b1 : avail_b - 8u;
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) {
                const uint2 lo = *reinterpret_cast<const uint2*>(rows_base + ro[rt] + b0);
                const uint2 hi = *reinterpret_cast<const uint2*>(rows_base + ro[rt] + b1);
                piece[rt][0] = lo.x;
                piece[rt][1] = lo.y;
                piece[rt][2] = hi.x;
                piece[rt][3] = hi.y;
            }
            narrow_group(piece, nlds + (size_t)g * 4u * 128u + lane * 2u, acc);
        }
        narrow_test(a, lcols, L, acc, lane, rbase, mac_any, span, tested_local);
    }
    __syncthreads();
    narrow_flush_counts(a, L);
    if (a.tested) {
        uint32_t v = tested_local;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) v += __shfl_xor(v, dd);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// Rows staged through LDS: a wave's 64 rows are one contiguous piece of memory (64 x stride bytes), fetched lane-linear
// (every load instruction reads 1 KB of consecutive bytes: each cache line is requested once, by one instruction), held
// in registers while the previous 64 rows are processed, written to the wave's own LDS area and read back from there
// as the 16-byte pieces the MFMA lanes need. NP = 16-byte pieces per lane = ceil(64 * stride / 1024). The launcher
// guarantees that every pass that exists can be read in whole KB without leaving the launch's rows.
// A row is read ONCE per scan: non-temporal loads (`global_load_dwordx4 ... nt`) keep the stream out of the L2's and the
// MALL's retention policy, and the memory system then delivers 7.1 TB/s to a pure read kernel instead of 6.0-6.3
// (tools/probe_hbm_read.hip: 8 KB per wave, lane-linear, as here). KGWAS_NARROW_NT=0 at compile time: plain loads.
// (Four blocks per CU instead of three - one column's operand kept as its 4 distinct rows, 4 KB instead of 16, constants out
// of LDS, 124 registers - measured SLOWER: 29.6 against 26.1-26.8 ms per 1.2 G rows. More waves are not what this kernel lacks.)
// (Persistent blocks - a grid of what the chip holds, each block walking row blocks b, b + grid, ... so that the operands
// are loaded once per block - measured SLOWER than one block per 1280 rows handed out by the hardware: 26.8-27.6 ms per
// 1.2 G rows against 26.3 at every grid size from 512 to 4096.)
#ifndef KGWAS_NARROW_NT
#define KGWAS_NARROW_NT 1
#endif
// KGWAS_NARROW_DEPTH: passes of rows in flight per wave. 1: the next pass. 2: the next two in a second register set -
// 176 registers with it, two waves per SIMD instead of three, 30.4 ms per 1.2 G rows instead of 26.1: measured, not used.
#ifndef KGWAS_NARROW_DEPTH
#define KGWAS_NARROW_DEPTH 1
#endif
__device__ __forceinline__ uint4 load_row_piece(const char* p) {
#if KGWAS_NARROW_NT
    typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
    const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

template <int NP>
__global__ void __launch_bounds__(256) narrow_staged_kernel(NarrowArgs a, uint32_t rows_per_block, uint32_t stage_bytes) {
    extern __shared__ uint4 nlds[];  // slice operands [n_steps][64][2], column constants, then per wave stage_bytes of rows
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // an SGPR: pass addresses are scalar
    const uint32_t kb = lane >> 4, m = lane & 15u;
    const uint32_t n_steps = a.n_kgroups * 4u;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.Bn);
        for (uint32_t i = threadIdx.x; i < n_steps * 128u; i += blockDim.x) nlds[i] = src[i];
    }
    NarrowCol* lcols = reinterpret_cast<NarrowCol*>(nlds + n_steps * 128u);
    if (threadIdx.x < a.n_pheno * (sizeof(NarrowCol) / 8u))
        reinterpret_cast<double*>(lcols)[threadIdx.x] = reinterpret_cast<const double*>(a.cols)[threadIdx.x];
    char* stage = reinterpret_cast<char*>(lcols) + 4u * sizeof(NarrowCol) + (size_t)wave * stage_bytes;
    __shared__ uint32_t lcnt_s[2 * NARROW_MAX_COLS];
    if (threadIdx.x < 2u * NARROW_MAX_COLS) lcnt_s[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t blk_row0 = (uint64_t)blockIdx.x * rows_per_block;
    const NarrowLane L = narrow_lane_constants(a, lcols, lane, lcnt_s, blk_row0);
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t stride_b = (uint32_t)a.src.stride_dw * 4u, off_b = a.src.off_dw * 4u;
    const uint32_t avail_b = a.src.avail_dw * 4u;
    const uint64_t total_b = a.n_rows * (uint64_t)stride_b;  // bytes of this launch's rows
    uint32_t tested_local = 0;
    const bool mac_any = a.S >= 2u * a.min_count;
    const uint32_t span = a.S - 2u * a.min_count;
    auto pass_row0 = [&](uint32_t ps) { return blk_row0 + (uint64_t)(ps * 4u + wave) * 64u; };
    auto pass_exists = [&](uint32_t ps) { return (ps * 4u + wave) * 64u < rows_per_block && pass_row0(ps) < a.n_rows; };
    // R: the NEXT pass's rows, lane-linear 16-byte pieces (named scalars, not an array: the compiler keeps a
    // loop-carried uint4 array in scratch memory)
    uint4 R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11;
#define NARROW_EACH(X) X(0, R0) X(1, R1) X(2, R2) X(3, R3) X(4, R4) X(5, R5) X(6, R6) X(7, R7) X(8, R8) X(9, R9) X(10, R10) X(11, R11)
#if KGWAS_NARROW_DEPTH == 2
    // ... and a second set: the rows of the pass after the next (the bytes in flight are what sets the rate of a read
    // stream whose latency under load is 4 us: 12 waves x 8.7 KB per CU keep 27 MB in flight on the chip, 6.2 TB/s)
    uint4 S0, S1, S2, S3, S4, S5, S6, S7, S8, S9, S10, S11;
#define NARROW_EACH2(X) X(0, S0) X(1, S1) X(2, S2) X(3, S3) X(4, S4) X(5, S5) X(6, S6) X(7, S7) X(8, S8) X(9, S9) X(10, S10) X(11, S11)
#endif
#define NARROW_LOAD(i, r) if (NP > i) r = (KGWAS_NARROW_ABLATE & 32) ? make_uint4(lane, i, (uint32_t)fb, 7u) : load_row_piece(fetch_ptr + 1024 * i);
#define NARROW_PUT(i, r) if (NP > i) *reinterpret_cast<uint4*>(stage + 16u * (64u * i + lane)) = r;
#define NARROW_FETCH(EACH, q)                                                                     \
    {                                                                                             \
        const uint64_t fb = pass_row0(q) * stride_b;                                              \
        const char* fetch_ptr = rows_base + fb + 16u * lane;                                      \
        if (fb < total_b && ((q) * 4u + wave) * 64u < rows_per_block) { /* (scalar) */            \
            EACH(NARROW_LOAD)                                                                     \
        }                                                                                         \
    }
    NARROW_FETCH(NARROW_EACH, 0u)
#if KGWAS_NARROW_DEPTH == 2
    NARROW_FETCH(NARROW_EACH2, 1u)
#endif
    // where this lane's pieces of the four row tiles sit in the wave's stage
    uint32_t pofs[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) pofs[rt] = (rt * 16u + m) * stride_b + off_b;
    // One pass: the wave's 64 rows from the register set EACH into its stage, the set refilled with the rows of pass
    // ps + KGWAS_NARROW_DEPTH (in flight while these rows are processed), MFMAs, test.
    auto process = [&](uint64_t rbase) {
        __builtin_amdgcn_wave_barrier();
        nv4f acc[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; rt++) acc[rt] = (nv4f){0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t g = 0; g < a.n_kgroups; g++) {
            uint32_t b0 = 64u * g + 16u * kb, b1 = b0 + 8u;
            b0 = b0 + 8u <= avail_b ? b0 : avail_b - 8u;
            b1 = b1 + 8u <= avail_b ? b1 : avail_b - 8u;
            uint32_t piece[NRT][4];
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) {
                const char* rp = stage + pofs[rt];
                uint2 lo, hi;
                if (KGWAS_NARROW_ABLATE & 8) {
                    lo = make_uint2(lane * 2654435761u + g, rt + b0);
                    hi = make_uint2(lane + b1, lane * 40503u);
                } else {
                    lo = *reinterpret_cast<const uint2*>(rp + b0);
                    hi = *reinterpret_cast<const uint2*>(rp + b1);
                }
                piece[rt][0] = lo.x;
                piece[rt][1] = lo.y;
                piece[rt][2] = hi.x;
                piece[rt][3] = hi.y;
            }
            narrow_group(piece, nlds + (size_t)g * 4u * 128u + lane * 2u, acc);
        }
        if (KGWAS_NARROW_ABLATE & 4) {
            float x = 0;
#pragma unroll
            for (int rt = 0; rt < NRT; rt++) x += acc[rt][0] + acc[rt][1] + acc[rt][2] + acc[rt][3];
            if (x == 12345.678f) tested_local++;
        } else {
            narrow_test(a, lcols, L, acc, lane, rbase, mac_any, span, tested_local);
        }
        __builtin_amdgcn_wave_barrier();  // the pass's piece reads are done before the next pass's rows are stored
    };
#define NARROW_PASS(EACH, ps)                                                                                     \
    {                                                                                                             \
        EACH(NARROW_PUT) /* (the last piece may reach past the 64 rows: the stage area is sized in whole KB) */   \
        NARROW_FETCH(EACH, (ps) + KGWAS_NARROW_DEPTH)                                                             \
        process(pass_row0(ps));                                                                                   \
    }
#if KGWAS_NARROW_DEPTH == 2
    for (uint32_t ps = 0;; ps += 2u) {
        if (!pass_exists(ps)) break;
        NARROW_PASS(NARROW_EACH, ps)
        if (!pass_exists(ps + 1u)) break;
        NARROW_PASS(NARROW_EACH2, ps + 1u)
    }
#undef NARROW_EACH2
#else
    for (uint32_t ps = 0; pass_exists(ps); ps++) NARROW_PASS(NARROW_EACH, ps)
#endif
#undef NARROW_PASS
#undef NARROW_FETCH
#undef NARROW_EACH
#undef NARROW_LOAD
#undef NARROW_PUT
    __syncthreads();
    narrow_flush_counts(a, L);
    if (a.tested) {
        uint32_t v = tested_local;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) v += __shfl_xor(v, dd);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// ---- bitmap -> row-ordered survivor keys, column by column (no sort) ------------------------------------------------
// Blocks of 1024 words (65 536 rows): counts, a scan of the block counts per column, then every block writes its keys
// (column << row_bits | row) at its offset, ascending rows: keys_sorted, surv_off[p], surv_cnt[p], key_count.
constexpr uint32_t BM_WORDS = 1024;

__global__ void __launch_bounds__(256) bitmap_count_kernel(const unsigned long long* bm, uint64_t words_per_col, uint32_t n_words,
                                                           uint32_t n_blocks, uint32_t* blk_cnt, unsigned long long* blk_mask) {
    __shared__ uint32_t part[4];
    const uint32_t p = blockIdx.y, b = blockIdx.x;
    const unsigned long long* w = bm + (uint64_t)p * words_per_col;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t x = b * BM_WORDS + threadIdx.x * 4u + i;
        if (x < n_words) c += __popcll(w[x]);
    }
    // which threads' words hold anything: the scatter kernel reads (and clears) only those - a steady chunk has a survivor
    // in one word of thirty, so its pass over the bitmap touches an eighth of the lines and the chunk's prep launch has
    // nothing to zero
    const unsigned long long m = __ballot(c != 0u);
    if ((threadIdx.x & 63u) == 0u) blk_mask[((uint64_t)p * n_blocks + b) * 4u + (threadIdx.x >> 6)] = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[p * n_blocks + b] = part[0] + part[1] + part[2] + part[3];
}

// block p: exclusive scan of column p's block counts (in place) and the column's total
__global__ void __launch_bounds__(256) bitmap_scan_kernel(uint32_t* blk_cnt, uint32_t n_blocks, uint32_t* col_total) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    const uint32_t p = blockIdx.x;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256u) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < n_blocks ? blk_cnt[p * n_blocks + b] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256u; d <<= 1) {
            const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
            __syncthreads();
            part[threadIdx.x] += x;
            __syncthreads();
        }
        if (b < n_blocks) blk_cnt[p * n_blocks + b] = carry + part[threadIdx.x] - v;  // offset inside the column's range
        __syncthreads();
        if (threadIdx.x == 255u) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) col_total[p] = carry;
}

// each column's range in the key list and the total: the work of one 256-thread block (block (0, 0) of the scatter launch - a
// launch of its own for these few microseconds cost the stream as much again)
__device__ __forceinline__ void bitmap_bases_block(const uint32_t* col_total, uint32_t n_pheno, uint32_t key_cap, uint32_t* surv_off,
                                                   uint32_t* surv_cnt, uint32_t* key_count, uint32_t* tile_pref) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < n_pheno; p0 += 256u) {
        const uint32_t p = p0 + threadIdx.x;
        const uint32_t v = p < n_pheno ? col_total[p] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256u; d <<= 1) {
            const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
            __syncthreads();
            part[threadIdx.x] += x;
            __syncthreads();
        }
        if (p < n_pheno) {
            surv_off[p] = carry + part[threadIdx.x] - v;
            surv_cnt[p] = v;
        }
        __syncthreads();
        if (threadIdx.x == 255u) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *key_count = carry;  // the host compares it with the capacity
    const bool over = carry > key_cap;  // overflow: the chunk is redone in halves; leave nothing for the re-score kernels to walk
    __syncthreads();
    if (over)
        for (uint32_t p = threadIdx.x; p < n_pheno; p += 256u) surv_cnt[p] = 0u;
    // the re-score kernels' tiles of 256 survivors, never straddling a column: tile_pref[p] = first tile of column p
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t p0 = 0; p0 < n_pheno; p0 += 256u) {
        const uint32_t p = p0 + threadIdx.x;
        const uint32_t v = (p < n_pheno && !over) ? (col_total[p] + 255u) / 256u : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 256u; d <<= 1) {
            const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
            __syncthreads();
            part[threadIdx.x] += x;
            __syncthreads();
        }
        if (p < n_pheno) tile_pref[p] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255u) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_pref[n_pheno] = carry;
    __syncthreads();
}

// blk_off: the scanned block counts (bitmap_scan_kernel), col_total: the columns' totals - a block's own count is the
// difference to the next offset. Only the threads the count kernel marked load their words, and they store zeros back: the
// bitmap is all zero again when the launch ends (scan_gpu.cpp: bitmap_clean), whatever the key list could hold.
__global__ void __launch_bounds__(256) bitmap_scatter_kernel(unsigned long long* bm, uint64_t words_per_col, uint32_t n_words,
                                                             uint32_t n_blocks, const uint32_t* blk_off, const uint32_t* col_total,
                                                             const unsigned long long* blk_mask, uint32_t n_pheno, uint32_t* surv_off,
                                                             uint32_t* surv_cnt, uint32_t* key_count, uint32_t* tile_pref, uint32_t* keys,
                                                             uint32_t key_cap, uint32_t row_bits, bool nibble_transposed) {
    __shared__ uint32_t part[4];
    __shared__ uint32_t basep[4];
    const uint32_t p = blockIdx.y, b = blockIdx.x;
    // block (0, 0) writes what the kernels behind this one read: the columns' ranges, the total, the re-score tiles' table
    if (p == 0u && b == 0u) bitmap_bases_block(col_total, n_pheno, key_cap, surv_off, surv_cnt, key_count, tile_pref);
    const uint32_t my_off = blk_off[p * n_blocks + b];
    const uint32_t next_off = b + 1u < n_blocks ? blk_off[p * n_blocks + b + 1u] : col_total[p];
    if (next_off == my_off) return;  // nothing in this block (block-uniform)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // where column p's keys begin: the totals of the columns before it, added up by the block itself (surv_off[p] is block
    // (0, 0)'s to write, in this same launch)
    uint32_t col_base;
    {
        uint32_t v = 0;
        for (uint32_t q = threadIdx.x; q < p; q += 256u) v += col_total[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u) basep[wave] = v;
        __syncthreads();
        col_base = basep[0] + basep[1] + basep[2] + basep[3];
    }
    const bool mine = (blk_mask[((uint64_t)p * n_blocks + b) * 4u + wave] >> lane) & 1ull;
    unsigned long long* w = bm + (uint64_t)p * words_per_col;
    unsigned long long x[4] = {0ull, 0ull, 0ull, 0ull};
    uint32_t c = 0;
    if (mine) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t idx = b * BM_WORDS + threadIdx.x * 4u + i;
            if (idx < n_words) {
                x[i] = w[idx];
                if (x[i]) w[idx] = 0ull;
            }
            c += __popcll(x[i]);
        }
        if (nibble_transposed) {  // the int8 filters write nibble 4 kg + rt for rows 16 rt + 4 kg ..+3: back to row order
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (!x[i]) continue;
                unsigned long long y = 0;
#pragma unroll
                for (int kg = 0; kg < 4; kg++)
#pragma unroll
                    for (int rt = 0; rt < 4; rt++) y |= ((x[i] >> (4 * (4 * kg + rt))) & 0xFull) << (4 * (4 * rt + kg));
                x[i] = y;
            }
        }
    }
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63u) part[wave] = incl;
    __syncthreads();
    uint32_t o = col_base + my_off + incl - c;
    for (uint32_t k = 0; k < wave; k++) o += part[k];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned long long v = x[i];
        const uint32_t row0 = (b * BM_WORDS + threadIdx.x * 4u + i) * 64u;
        while (v) {
            const uint32_t bit = __ffsll((long long)v) - 1u;
            v &= v - 1ull;
            if (o < key_cap) keys[o] = (p << row_bits) | (row0 + bit);
            o++;
        }
    }
}

hipError_t launch_bitmap_keys(unsigned long long* bitmap, uint64_t words_per_col, uint64_t n_rows, uint32_t n_pheno, uint32_t* blk_scratch,
                              unsigned long long* blk_mask, uint32_t* keys_sorted, uint32_t key_cap, uint32_t row_bits, uint32_t* surv_off,
                              uint32_t* surv_cnt, uint32_t* key_count, uint32_t* tile_pref, bool nibble_transposed, hipStream_t st) {
    const uint32_t n_words = (uint32_t)((n_rows + 63) / 64);
    const uint32_t n_blocks = (n_words + BM_WORDS - 1) / BM_WORDS;
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(n_blocks, n_pheno), dim3(256), 0, st, bitmap, words_per_col, n_words, n_blocks, blk_scratch, blk_mask);
    uint32_t* col_total = blk_scratch + (size_t)n_pheno * n_blocks;
    hipLaunchKernelGGL(bitmap_scan_kernel, dim3(n_pheno), dim3(256), 0, st, blk_scratch, n_blocks, col_total);
    hipLaunchKernelGGL(bitmap_scatter_kernel, dim3(n_blocks, n_pheno), dim3(256), 0, st, bitmap, words_per_col, n_words, n_blocks, blk_scratch,
                       col_total, blk_mask, n_pheno, surv_off, surv_cnt, key_count, tile_pref, keys_sorted, key_cap, row_bits, nibble_transposed);
    return hipGetLastError();
}

// The narrow filter's keys in one launch: block (b, p) = segment b (1024 bitmap words) of column p. The survivors per
// (column, segment) are already counted (NarrowArgs::seg_cnt), so a block's place in the key list - everything of the
// columns before it and of its own column's earlier segments - is a sum over at most a few thousand counters that every
// block forms for itself: no scan kernel, no inter-block dependency. Block (0, 0) also writes the columns' ranges, the
// re-score tiles' table and meta.
__global__ void __launch_bounds__(256) narrow_keys_kernel(const unsigned long long* bm, uint64_t words_per_col, uint32_t n_words, uint32_t n_segs,
                                                          uint32_t n_pheno, const uint32_t* seg_cnt, uint32_t* keys, uint32_t key_cap,
                                                          uint32_t row_bits, uint32_t* surv_off, uint32_t* surv_cnt, uint32_t* key_count,
                                                          uint32_t* tile_pref, uint32_t* meta, const unsigned long long* tested_shards) {
    __shared__ uint32_t red[4][NARROW_MAX_COLS + 1];
    __shared__ uint32_t tot[NARROW_MAX_COLS + 1];
    __shared__ uint32_t part[4];
    const uint32_t p = blockIdx.y, b = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // col_sum[q] = survivors of column q; mine = those of column p's segments before b
    uint32_t cs[NARROW_MAX_COLS] = {0u, 0u, 0u, 0u}, mine = 0;
    for (uint32_t q = 0; q < n_pheno; q++)
        for (uint32_t k = threadIdx.x; k < n_segs; k += 256u) {
            const uint32_t v = seg_cnt[q * n_segs + k];
            cs[q] += v;
            if (q == p && k < b) mine += v;
        }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        mine += __shfl_xor(mine, d);
#pragma unroll
        for (int q = 0; q < (int)NARROW_MAX_COLS; q++) cs[q] += __shfl_xor(cs[q], d);
    }
    if (lane == 0u) {
#pragma unroll
        for (int q = 0; q < (int)NARROW_MAX_COLS; q++) red[wave][q] = cs[q];
        red[wave][NARROW_MAX_COLS] = mine;
    }
    __syncthreads();
    if (threadIdx.x <= NARROW_MAX_COLS) tot[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
    uint32_t total = 0, off_p = 0;
    for (uint32_t q = 0; q < n_pheno; q++) {
        if (q < p) off_p += tot[q];
        total += tot[q];
    }
    const bool over = total > key_cap;  // the chunk is redone in halves: nothing for the re-score kernel to walk
    if (b == 0u && p == 0u) {  // the filter's MAC-passing-row counters -> meta[2 P + 2 .. 3]: one copy closes the chunk
        __shared__ unsigned long long tsum[4];
        unsigned long long v = tested_shards[threadIdx.x];  // TESTED_SHARDS == 256 == blockDim.x
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u) tsum[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0u) {
            const unsigned long long tt = tsum[0] + tsum[1] + tsum[2] + tsum[3];
            meta[2u * n_pheno + 2u] = (uint32_t)tt;
            meta[2u * n_pheno + 3u] = (uint32_t)(tt >> 32);
        }
    }
    if (b == 0u && p == 0u && threadIdx.x == 0u) {
        uint32_t o = 0, t = 0;
        for (uint32_t q = 0; q < n_pheno; q++) {
            const uint32_t c = over ? 0u : tot[q];
            surv_off[q] = o;
            surv_cnt[q] = c;
            tile_pref[q] = t;
            meta[q] = c;             // records of column q: its survivors (a survivor that is no candidate carries -inf)
            meta[n_pheno + q] = o;
            o += tot[q];
            t += (c + 255u) / 256u;
        }
        tile_pref[n_pheno] = t;
        *key_count = total;
        meta[2u * n_pheno] = over ? 0u : total;
        meta[2u * n_pheno + 1u] = total;
    }
    if (over || seg_cnt[p * n_segs + b] == 0u) return;  // block-uniform
    const unsigned long long* w = bm + (uint64_t)p * words_per_col;
    unsigned long long x[4];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t idx = b * BM_WORDS + threadIdx.x * 4u + i;
        x[i] = idx < n_words ? w[idx] : 0ull;
        c += __popcll(x[i]);
    }
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((int)lane >= d) incl += t;
    }
    if (lane == 63u) part[wave] = incl;
    __syncthreads();
    uint32_t o = off_p + tot[NARROW_MAX_COLS] + incl - c;
    for (uint32_t k = 0; k < wave; k++) o += part[k];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned long long v = x[i];
        const uint32_t row0 = (b * BM_WORDS + threadIdx.x * 4u + i) * 64u;
        while (v) {
            const uint32_t bit = __ffsll((long long)v) - 1u;
            v &= v - 1ull;
            if (o < key_cap) keys[o] = (p << row_bits) | (row0 + bit);
            o++;
        }
    }
}

hipError_t launch_narrow_keys(const unsigned long long* bitmap, uint64_t words_per_col, uint64_t n_rows, uint32_t n_pheno, const uint32_t* seg_cnt,
                              uint32_t* keys_sorted, uint32_t key_cap, uint32_t row_bits, uint32_t* surv_off, uint32_t* surv_cnt,
                              uint32_t* key_count, uint32_t* tile_pref, uint32_t* meta, const unsigned long long* tested_shards, hipStream_t st) {
    static_assert(TESTED_SHARDS == 256, "block (0, 0) sums one counter per thread");
    if (n_pheno < 1 || n_pheno > NARROW_MAX_COLS) return hipErrorInvalidValue;
    const uint32_t n_words = (uint32_t)((n_rows + 63) / 64);
    const uint32_t n_segs = (n_words + BM_WORDS - 1) / BM_WORDS;
    hipLaunchKernelGGL(narrow_keys_kernel, dim3(n_segs, n_pheno), dim3(256), 0, st, bitmap, words_per_col, n_words, n_segs, n_pheno, seg_cnt,
                       keys_sorted, key_cap, row_bits, surv_off, surv_cnt, key_count, tile_pref, meta, tested_shards);
    return hipGetLastError();
}

size_t narrow_lds_bytes(uint32_t n_kgroups) { return (size_t)n_kgroups * 4u * 2048u + 4u * sizeof(NarrowCol); }

template <int NP>
static hipError_t launch_staged_t(const NarrowArgs& a, uint32_t rows_per_block, uint32_t n_blocks, uint32_t stage_bytes, size_t lds,
                                  hipStream_t st, bool last) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)narrow_staged_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (last)  // (no second launch for a tail behind it: a caller's event is bound to this one, launch.h)
        launch_last(narrow_staged_kernel<NP>, dim3(n_blocks), dim3(256), lds, st, a, rows_per_block, stage_bytes);
    else
        hipLaunchKernelGGL((narrow_staged_kernel<NP>), dim3(n_blocks), dim3(256), lds, st, a, rows_per_block, stage_bytes);
    return hipGetLastError();
}

hipError_t launch_narrow(const NarrowArgs& a, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    if (a.n_pheno < 1 || a.n_pheno > NARROW_MAX_COLS) return hipErrorInvalidValue;
    rows_per_block = (rows_per_block + 255u) / 256u * 256u;
    // staged rows: 64 x stride bytes per wave beside the operands, three blocks per CU; the rows must start on a 16-byte
    // boundary
    const uint32_t stride_b = (uint32_t)a.src.stride_dw * 4u;
    const int np = (int)((64u * stride_b + 1023u) / 1024u);
    const uint32_t stage_bytes = (uint32_t)np * 1024u;  // whole lane-linear pieces
    const size_t lds_staged = narrow_lds_bytes(a.n_kgroups) + 4u * (size_t)stage_bytes;
    static const bool no_stage = exp_int("KGWAS_NARROW_STAGED", 1) == 0;  // experiments
    // The staged kernel reads a pass (64 rows) as np whole KB, i.e. up to 1 KB - 16 past the 64 rows: it takes the
    // rows whose passes can be read that way without leaving the launch's rows; the last few rows (fewer than 128 + a
    // KB's worth) go through the direct kernel in a second, tiny launch.
    uint64_t n_staged = 0;
    if (!no_stage && lds_staged <= 53u * 1024u && np <= 12 && (reinterpret_cast<uintptr_t>(a.src.base) & 15u) == 0) {
        // All rows, if the last pass (whole or not: rows past n_rows fail narrow_test's row test) can be read in whole KB
        // inside the buffer - a.slack_rows rows of the same buffer follow the launch's rows, which every chunk of a feed
        // but the last has. Otherwise the passes that can, and a second launch for the rest.
        const uint64_t readable_b = (a.n_rows + a.slack_rows) * (uint64_t)stride_b;
        const uint64_t total_b = a.n_rows * (uint64_t)stride_b, over = (uint64_t)np * 1024u - 64ull * stride_b;
        if ((a.n_rows - 1u) / 64u * 64u * stride_b + (uint64_t)np * 1024u <= readable_b)
            n_staged = a.n_rows;
        else if (total_b > over)
            n_staged = (total_b - over) / (64ull * stride_b) * 64ull;
    }
    if (n_staged) {
        NarrowArgs s1 = a;
        s1.n_rows = n_staged;
        const uint32_t nb = (uint32_t)((n_staged + rows_per_block - 1) / rows_per_block);
        hipError_t e = hipSuccess;
        switch (np) {
            case 1: e = launch_staged_t<1>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 2: e = launch_staged_t<2>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 3: e = launch_staged_t<3>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 4: e = launch_staged_t<4>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 5: e = launch_staged_t<5>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 6: e = launch_staged_t<6>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 7: e = launch_staged_t<7>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 8: e = launch_staged_t<8>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 9: e = launch_staged_t<9>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 10: e = launch_staged_t<10>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            case 11: e = launch_staged_t<11>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
            default: e = launch_staged_t<12>(s1, rows_per_block, nb, stage_bytes, lds_staged, st, n_staged == a.n_rows); break;
        }
        if (e != hipSuccess || n_staged == a.n_rows) return e;
    }
    NarrowArgs a2 = a;  // what is left (everything, if the staged kernel does not apply)
    a2.src.base = a.src.base + n_staged * a.src.stride_dw;
    a2.n_rows = a.n_rows - n_staged;
    a2.row_off = (uint32_t)n_staged;
    const uint32_t n_blocks2 = (uint32_t)((a2.n_rows + rows_per_block - 1) / rows_per_block);
    const size_t lds = narrow_lds_bytes(a.n_kgroups);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)narrow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    launch_last(narrow_kernel, dim3(n_blocks2), dim3(256), lds, st, a2, rows_per_block);
    return hipGetLastError();
}

}  // namespace kgwas
