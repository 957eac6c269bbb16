// kin_kernels.hip — emma_kinship_kmers on the GPU (src/emma_kinship_kmers.cpp:77-102,
// update_emma_kinshhip_calculation, src/kmers_multiple_databases.cpp:418-438).
//
// K[i][j] += 1 ^ g_i ^ g_j over the rows that pass ceil(S_f*maf) <= popcount <= S_f - that equals
// n_used - Hamming(i, j), and Hamming(i, j) = c_ii + c_jj - 2 c_ij with c_ij = sum_rows g_i g_j: the accumulation is
// ONE symmetric integer Gram matrix of the bit table, exact in int32 per chunk and u64 across chunks. Two kernels per
// chunk of rows:
//   kin_transpose_kernel : MAC filter over all S_f columns (failing rows contribute zero bits) + bit transpose into
//                          sample-major planes T[sample][row/32], 32x32 bit tiles transposed across 32 lanes in five
//                          butterfly stages (the previous version took one ballot per sample: 1152 per 64 rows,
//                          half of the whole accumulation's time);
//   kin_gram_kernel      : C = T T^t on gfx950's block-scaled MFMA with BOTH operands in FP4 (E2M1): a nibble with
//                          only bit 0, 1 or 2 set is 0.5, 1.0 or 2.0, so `plane dword & (0x11111111 << j)` is an
//                          operand of eight row bits in ONE lane-op, and the operands' block scales (2^(1-j), exact)
//                          make every product of two set bits exactly 1.0; bit 3 (the FP4 sign) is shifted first.
//                          v_mfma_scale_f32_16x16x128_f8f6f4 runs FP4 x FP4 at twice the int8 rate; float32
//                          accumulation of 0/1 products is exact below 2^24 rows per launch slice. A wave owns a
//                          64 x 128 tile of C (32 accumulators) and expands its own operands: 60 lane-ops per 128
//                          MFMAs' worth of operands per 512 rows (the int8 form this replaces: v_mfma_i32_16x16x64_i8,
//                          (dword >> j) & 0x01010101, 96 lane-ops per 32 MFMAs, 0.33 of the int8 peak). The
//                          xor-popcount formulation before that does S^2/64 lane-ops per row (VALU-bound, 146 T
//                          pair-updates/s at 1135 samples).
#include <stdlib.h>

#include <algorithm>

#include "kernels.h"

#ifndef KGWAS_KIN_ABLATE
#define KGWAS_KIN_ABLATE 0
#endif

namespace kgwas {

typedef int kin_i32x4 __attribute__((ext_vector_type(4)));
typedef int kin_i32x8 __attribute__((ext_vector_type(8)));
typedef float kin_f32x4 __attribute__((ext_vector_type(4)));

namespace {

// FP4 operand of MFMA step j = 0..3 from four plane dwords (128 rows of one sample): nibble e of operand dword q is
// bit 4e + j of plane dword q, i.e. k-element 8q + e <-> row 32q + 4e + j of the lane's 128 (A and B operands use the
// same map, so the k index pairs a row with itself). Steps 0..2 are one AND per dword (values 0.5, 1, 2: the block
// scale 2^(1-j) makes them 1), step 3 moves the FP4 sign bit down first (value 2, scale 2^-1).
__device__ __forceinline__ kin_i32x8 kin_expand_step(const uint4& w, int j) {
    kin_i32x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
    if (KGWAS_KIN_ABLATE & 32) {  // experiments: no expansion work
        r[0] = (int)w.x;
        r[1] = (int)w.y;
        r[2] = (int)w.z;
        r[3] = (int)w.w;
        return r;
    }
    if (j < 3) {
        const uint32_t mk = 0x11111111u << j;
        r[0] = (int)(w.x & mk);
        r[1] = (int)(w.y & mk);
        r[2] = (int)(w.z & mk);
        r[3] = (int)(w.w & mk);
    } else {
        r[0] = (int)((w.x >> 1) & 0x44444444u);
        r[1] = (int)((w.y >> 1) & 0x44444444u);
        r[2] = (int)((w.z >> 1) & 0x44444444u);
        r[3] = (int)((w.w >> 1) & 0x44444444u);
    }
    return r;
}

// x of lane (lane ^ J) for J = 1, 2, 4, 8 as DPP moves (vector-ALU operand modifiers: no trip through the LDS
// crossbar that __shfl_xor's ds_bpermute takes): quad permutes for 1 and 2; for 4 and 8 a row shift left into the
// lanes whose bit J is clear and a row shift right into the others, selected by the DPP bank mask (banks = groups
// of four lanes of a 16-lane row).
template <int J>
__device__ __forceinline__ uint32_t lane_xor_dpp(uint32_t x) {
    int r = (int)x;
    if (J == 1) {
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    } else if (J == 2) {
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    } else if (J == 4) {
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0, 2
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xF, 0xA, false);  // row_shr:4 into banks 1, 3
    } else {
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x108, 0xF, 0x3, false);  // row_shl:8 into banks 0, 1
        r = __builtin_amdgcn_update_dpp(r, (int)x, 0x118, 0xF, 0xC, false);  // row_shr:8 into banks 2, 3
    }
    return (uint32_t)r;
}

// 32 x 32 bit-matrix transpose across the 32 lanes of a half wave: in: lane r holds row r (bit s = sample s);
// out: lane s holds sample s (bit r = row r). Five butterfly stages; only the first (lanes 16 apart) crosses a
// 16-lane row and goes through the LDS crossbar.
template <int J>
__device__ __forceinline__ void transpose_stage(uint32_t& x, uint32_t& m, uint32_t lane) {
    uint32_t y;
    if (J == 16) {
        // x of lane ^ 16 without the LDS crossbar: v_permlane16_swap_b32 exchanges the odd 16-lane rows of its first
        // operand with the even rows of its second; with x as both, the first result holds rows (0, 0, 2, 2) of x and
        // the second rows (1, 1, 3, 3)
        const auto sw = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        y = (lane & 16u) ? sw[0] : sw[1];
    } else {
        y = lane_xor_dpp<J>(x);
    }
    const bool lo = (lane & (uint32_t)J) == 0u;
    const uint32_t a = lo ? x : y, b = lo ? y : x;
    const uint32_t t = ((a >> J) ^ b) & m;
    x = lo ? (a ^ (t << J)) : (b ^ t);
    m ^= m << (J >> 1);
}
__device__ __forceinline__ uint32_t transpose32(uint32_t x, uint32_t lane) {
    uint32_t m = 0x0000FFFFu;
    transpose_stage<16>(x, m, lane);
    transpose_stage<8>(x, m, lane);
    transpose_stage<4>(x, m, lane);
    transpose_stage<2>(x, m, lane);
    transpose_stage<1>(x, m, lane);
    return x;
}

}  // namespace

// Plane word rw (u32) of sample c = its presence bits for rows 32*rw .. 32*rw+31 of the launch, stored tile-major as
// T[rw / 8][c][rw % 8]; rows failing the filter are all-zero. A block covers blockDim.x = 256 rows = one tile of 8
// plane words (76 KB of LDS at 1135 samples: two blocks per CU, so one's loads overlap the other's transposes); where
// 256 verbatim rows plus their planes do not fit the LDS (more than ~2500 accessions) a block takes 128 or 64 rows,
// i.e. 4 or 2 of a tile's 8 plane words (launch_kin_transpose).
// Two threads per row (blockDim.x = 2 x rows per block): thread (row, half h) counts and transposes half of the row's
// dwords, so a block's phases - copy in, count, transpose, copy out - run on twice the waves for the same LDS footprint
// (the footprint, not the registers, is what limits a CU to two of these blocks).
// DIRECT_IN (experiments, launch_kin_transpose: measured slower): no verbatim copy of the rows in LDS - every thread loads its
// own part of its row (at most KIN_DIRECT_DW dwords) from global memory into registers, once, for the count and for the
// transposes: the block's LDS is the planes alone (37 KB instead of 76 at 1135 samples), one barrier and two passes over LDS
// fewer; but a wave's loads touch 64 rows x 36 bytes at a 152-byte stride - 64 addresses per instruction.
constexpr uint32_t KIN_DIRECT_DW = 12u;
#ifndef KGWAS_KIN_FLIP
#define KGWAS_KIN_FLIP 1  // rows with more than half of their bits set enter the planes complemented (kin_transpose_kernel)
#endif
template <bool DIRECT_IN>
__global__ void __launch_bounds__(1024) kin_transpose_kernel(const uint64_t* file_rows, uint64_t file_stride_w, uint64_t n_rows,
                                                            uint32_t S_f, uint32_t S_pad, uint32_t min_count, uint32_t* T,
                                                            uint64_t n_rw, unsigned long long* n_used, uint32_t rpb) {
    extern __shared__ uint32_t kin_lds[];
    const uint32_t tpr = blockDim.x / rpb;  // threads per row (rows per block: 256, 128 or 64)
    const uint32_t wpb = rpb / 32u;        // plane words per block and sample
    const uint32_t stride_dw = (uint32_t)(2u * file_stride_w);
    uint32_t* lin = kin_lds;                                            // [rpb][stride_dw] verbatim rows (k-mer word included)
    uint32_t* lout = kin_lds + (DIRECT_IN ? 0u : rpb * stride_dw);     // [S_pad][wpb]
    uint32_t* pn = lout + S_pad * wpb;                                  // [tpr][rpb] partial popcounts of the parts of a row
    const uint32_t in_dw = 2u * ((S_f + 63u) / 64u);
    const uint64_t row0 = (uint64_t)blockIdx.x * rpb;
    const uint32_t rr = threadIdx.x % rpb, half = threadIdx.x / rpb;
    const uint32_t lane = threadIdx.x & 63u, wave = rr >> 6;  // wave: of the row set
    const uint64_t r = row0 + rr;
    const uint32_t n_d = S_pad / 32u, d_half = (n_d + tpr - 1u) / tpr;
    const uint32_t d0 = half * d_half < n_d ? half * d_half : n_d, d1 = (d0 + d_half < n_d) ? d0 + d_half : n_d;
    uint32_t own[KIN_DIRECT_DW];
    if (DIRECT_IN) {
        const uint32_t* g = reinterpret_cast<const uint32_t*>(file_rows + (r < n_rows ? r : 0) * file_stride_w) + 2u;
#pragma unroll
        for (uint32_t i = 0; i < KIN_DIRECT_DW; i++) own[i] = (r < n_rows && d0 + i < d1 && d0 + i < in_dw) ? g[d0 + i] : 0u;
    } else if (row0 < n_rows) {  // coalesced verbatim copy of up to rpb contiguous rows
        const uint64_t left = n_rows - row0;
        const uint32_t n2 = (uint32_t)((left < rpb ? left : rpb) * file_stride_w);
        const uint2* src = reinterpret_cast<const uint2*>(file_rows + row0 * file_stride_w);
        uint2* dst = reinterpret_cast<uint2*>(lin);
        for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) dst[i] = src[i];
    }
    if (!DIRECT_IN) __syncthreads();
    const uint32_t* my = lin + (size_t)rr * stride_dw + 2u;
    // file padding bits (>= S_f) count neither in the predicate nor in the planes: calculate_unsqueezed_popcnt masks
    // with m_map_mask (src/kmers_multiple_databases.cpp:149-154)
    auto mask_of = [&](uint32_t d, uint32_t x) {
        if (32u * d + 32u > S_f) x &= (32u * d < S_f) ? ((1u << (S_f - 32u * d)) - 1u) : 0u;
        return x;
    };
    auto masked = [&](uint32_t d) { return mask_of(d, my[d]); };
    if (DIRECT_IN) {
#pragma unroll
        for (uint32_t i = 0; i < KIN_DIRECT_DW; i++) own[i] = mask_of(d0 + i, own[i]);
    }
    {
        uint32_t c = 0;
        if (DIRECT_IN) {
#pragma unroll
            for (uint32_t i = 0; i < KIN_DIRECT_DW; i++) c += __popc(own[i]);  // (zero beyond the thread's part and beyond the rows)
        } else if (r < n_rows) {
            for (uint32_t d = d0; d < d1 && d < in_dw; d++) c += __popc(masked(d));
        }
        pn[half * rpb + rr] = c;
    }
    __syncthreads();
    uint32_t n1 = 0;
    for (uint32_t q = 0; q < tpr; q++) n1 += pn[q * rpb + rr];
    // src/emma_kinship_kmers.cpp:83,89 -> load_kmers' predicate with all S_f columns
    const bool pass = (r < n_rows) && (S_f >= min_count) && (n1 >= min_count) && (n1 <= S_f - min_count);
    const unsigned long long kept = half == 0u ? __popcll(__ballot(pass)) : 0ull;  // (half is wave-uniform: rpb >= 64)
#if KGWAS_KIN_FLIP
    // Hamming(i, j) = sum_rows g_i ^ g_j = c_ii + c_jj - 2 c_ij does not change when a ROW is complemented (both of its bits
    // flip), and that is all the Gram matrix is used for (kgwas_kinship_partials): a row with more than half of its S_f bits set
    // goes into the planes complemented (its padding bits stay zero). The planes' density falls from the rows' mean frequency to
    // the mean of min(f, 1 - f) - 0.5 -> 0.26 on the synthetic tables -, and with it the operands' toggle rate under the Gram
    // kernel's MFMAs: the board is power-limited on this instruction, so sparser operands are clock.
    const uint32_t flipm = (n1 * 2u > S_f) ? 0xFFFFFFFFu : 0u;
#else
    const uint32_t flipm = 0u;
#endif
    auto plane_store = [&](uint32_t d, uint32_t x) {
        x = transpose32(x, lane);
        // lane s of 32-lane group g now holds sample 32d + s over the group's 32 rows
        // (the word index is XORed with bits 3..5 of the sample where a sample has 8 plane words: 64 lanes, 64 banks)
        const uint32_t smp = 32u * d + (lane & 31u), wd = wave * 2u + (lane >> 5);
        lout[smp * wpb + (wpb == 8u ? (wd ^ ((smp >> 3) & 7u)) : wd)] = x;
    };
    if (DIRECT_IN) {
#pragma unroll
        for (uint32_t i = 0; i < KIN_DIRECT_DW; i++)
            if (d0 + i < d1) plane_store(d0 + i, pass ? mask_of(d0 + i, own[i] ^ flipm) : 0u);  // (d0, d1 are wave-uniform: transpose32 runs with all lanes)
    } else {
        for (uint32_t d = d0; d < d1; d++) plane_store(d, (pass && d < in_dw) ? mask_of(d, my[d] ^ flipm) : 0u);
    }
    if (lane == 0 && kept) atomicAdd(&n_used[blockIdx.x % TESTED_SHARDS], kept);  // each wave adds the rows it counted
    __syncthreads();
    if (rpb == 256u) {
        // this block's 256 rows of all samples are one contiguous 32*S_pad-byte piece; the Gram kernel's round (16
        // plane words of 128 samples) is two 4 KB pieces
        uint32_t* dst = T + (uint64_t)blockIdx.x * S_pad * 8u;
        for (uint32_t e = threadIdx.x; e < S_pad * 8u; e += blockDim.x) dst[e] = lout[(e & ~7u) + ((e & 7u) ^ ((e >> 6) & 7u))];
    } else {
        const uint32_t per_tile = 256u / rpb;  // blocks per tile of 8 plane words
        const uint64_t tile = blockIdx.x / per_tile;
        const uint32_t w0 = (blockIdx.x % per_tile) * wpb;
        uint32_t* dst = T + tile * S_pad * 8u + w0;  // sample c: dst[c * 8 + i], i < wpb
        for (uint32_t e = threadIdx.x; e < S_pad * wpb; e += blockDim.x) dst[(e / wpb) * 8u + (e % wpb)] = lout[e];
    }
}

// C[i][j] += sum over the launch's rows of g_i g_j for the 128 x 128 sample tile (ib, jb), jb >= ib, and a slice of
// the rows (blockIdx.y). Block = 2 waves; wave w owns samples ib*128 + 64w .. +63 against all 128 of jb.
// MFMA k index <-> rows: per round of 512 rows lane (m, kg) holds plane dwords 4kg .. 4kg+3 (rows 128kg .. +127) of
// its samples and step j = 0..7 takes bit j of each of their bytes (kin_expand_step).
#ifndef KGWAS_KIN_KC
#define KGWAS_KIN_KC 16
#endif
constexpr uint32_t KIN_KC = KGWAS_KIN_KC;  // plane dwords (512 rows) staged per round
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) kin_gram_kernel(const uint32_t* T, uint64_t n_rw, uint32_t S_pad, kin_f32x4* part,
                                                       uint64_t rw_per_split, uint32_t tiles) {
    // A wave's operands of one round, as its lanes will hold them: slot s (0..3 = its 4 x 16 A samples, 4..11 = the 8 x 16
    // B samples) x 64 lanes x 16 bytes. Wave-private: no block barrier anywhere in this kernel.
    __shared__ __attribute__((aligned(16))) uint4 stage[2][12][64];
    // Blocks go to XCD (block id % 8) in launch order; each XCD takes a contiguous range of (slice, tile) work items,
    // slice-major, so that the ~45 tiles of a row slice - which read the same plane words round after round - run beside
    // each other under ONE L2 (each 128-sample block of planes is an operand of nt + 1 tiles: ten times the planes'
    // size crosses the L2 per chunk; spread over all XCDs by the default order it all came from the MALL / HBM).
    uint32_t wg;
    {
        const uint32_t nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        wg = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const uint32_t split = wg / tiles, tile = wg % tiles;
    uint32_t tix = tile, ib = 0;  // upper-triangular tile index -> (ib, jb), jb >= ib
    const uint32_t nt = S_pad / 128u;
    while (tix >= nt - ib) {
        tix -= nt - ib;
        ib++;
    }
    const uint32_t jb = ib + tix;
    const uint64_t k_begin = (uint64_t)split * rw_per_split;
    uint64_t k_end = k_begin + rw_per_split;
    if (k_end > n_rw) k_end = n_rw;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t m = lane & 15u, kg = lane >> 4;
    kin_f32x4 acc[4][8];
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 8; y++) acc[x][y] = (kin_f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // A round is KIN_KC = 16 plane dwords (512 rows) of the wave's 64 + 128 samples. Lane (m, kg) multiplies dwords
    // 4 kg ..+3 of sample 16 s + m for each of its 12 operand slots s, and with the tile-major planes
    // T[8-dword group][sample][8] those 16 bytes are contiguous: every lane fetches exactly its own operands, slot by
    // slot, with `global_load_lds_dwordx4` (global -> LDS without registers, lane l's 16 bytes land at slot base +
    // 16 l) one round ahead, and picks them up with one ds_read_b128 per slot when the round starts. The LDS is only a
    // register-free landing zone here - nothing staged is shared, so the two waves of a block never wait for each other
    // (the version that shared the B samples through LDS spent a third of its time at two barriers per round). Slices
    // are whole rounds (rw_per_split and n_rw are multiples of KIN_KC): nothing is clamped or masked.
    static_assert(KIN_KC == 16, "one round = two 8-dword plane groups");
    const uint32_t offA = ((kg >> 1) * S_pad + ib * 128u + wave * 64u + m) * 8u + 4u * (kg & 1u);
    const uint32_t offB = ((kg >> 1) * S_pad + jb * 128u + m) * 8u + 4u * (kg & 1u);
    uint4* const wstage = &stage[wave][0][0];  // wave-uniform
    auto issue = [&](uint64_t k0) {
        const uint32_t* base = T + k0 * S_pad;  // wave-uniform
#pragma unroll
        for (int sl = 0; sl < 12; sl++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (sl < 4 ? offA + sl * 128u : offB + (sl - 4) * 128u)),
                                             (__attribute__((address_space(3))) void*)(wstage + sl * 64), 16, 0, 0);
    };
    if (k_begin < k_end) issue(k_begin);
    // (Two waves share a SIMD and walk their rounds in step; starting the odd wave slot half a round late changed
    // nothing. Decomposition at 8 M rows x 1135, transposes excluded: 2.7 ms, of which the MFMA stream alone 2.3 - the
    // chip runs this FP4 stream at ~1.7 GHz (GRBM_GUI_ACTIVE), not 2.4 -, operand reads 0.2, expansion 0.2.)
    uint4 wA[4], wB[8];
    for (uint64_t k0 = k_begin; k0 < k_end; k0 += KIN_KC) {
        __builtin_amdgcn_s_waitcnt(0);  // this round's operands have landed
        asm volatile("" ::: "memory");
        if (!(KGWAS_KIN_ABLATE & 64) || k0 == k_begin) {
#pragma unroll
            for (int x = 0; x < 4; x++) wA[x] = wstage[x * 64 + lane];
#pragma unroll
            for (int y = 0; y < 8; y++) wB[y] = wstage[(4 + y) * 64 + lane];
        }
        if (KGWAS_KIN_ABLATE & 64) {
#pragma unroll
            for (int x = 0; x < 4; x++) asm volatile("" : "+v"(wA[x].x), "+v"(wA[x].y), "+v"(wA[x].z), "+v"(wA[x].w));
#pragma unroll
            for (int y = 0; y < 8; y++) asm volatile("" : "+v"(wB[y].x), "+v"(wB[y].y), "+v"(wB[y].z), "+v"(wB[y].w));
        }
        if (KGWAS_KIN_ABLATE & 16) {  // experiments: same instruction stream on all-zero operands (power / clocks)
            uint32_t z = 0;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
            for (int x = 0; x < 4; x++) wA[x] = make_uint4(wA[x].x & z, wA[x].y & z, wA[x].z & z, wA[x].w & z);
#pragma unroll
            for (int y = 0; y < 8; y++) wB[y] = make_uint4(wB[y].x & z, wB[y].y & z, wB[y].z & z, wB[y].w & z);
        }
        if (k0 + KIN_KC < k_end && !(KGWAS_KIN_ABLATE & 2)) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0);  // ... and are in registers: the landing zone is free for the next round's
            issue(k0 + KIN_KC);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            kin_i32x8 A[4];
#pragma unroll
            for (int x = 0; x < 4; x++) A[x] = kin_expand_step(wA[x], j);
#pragma unroll
            for (int y = 0; y < 8; y++) {
                const kin_i32x8 B = kin_expand_step(wB[y], j);
                // both operands FP4 (cbsz = blgp = 4); block scales 2^1, 2^0, 2^-1, 2^-1 for j = 0..3 on both sides
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    if (KGWAS_KIN_ABLATE & 1) {  // experiments: operands kept alive, no matrix work
                        acc[x][y][0] += __int_as_float((A[x][0] ^ B[1]) & 1);
                        continue;
                    }
                    if (j == 0) acc[x][y] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[x], B, acc[x][y], 4, 4, 0, (int)0x80808080, 0, (int)0x80808080);
                    if (j == 1) acc[x][y] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[x], B, acc[x][y], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    if (j >= 2) acc[x][y] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[x], B, acc[x][y], 4, 4, 0, 0x7E7E7E7E, 0, 0x7E7E7E7E);
                }
            }
        }
    }
    // The slice's 128 x 128 partial goes to its own 64 KB piece of `part` as the accumulator fragments lie in the
    // registers (one 16-byte store per lane and 16 x 16 sub-tile, 1 KB per instruction); kin_reduce_kernel adds the
    // slices of a tile into C. (Per-element 64-bit atomicAdds - 33 M per 2^20-row chunk at 1135 samples - were a
    // quarter of this kernel's time.)
    kin_f32x4* out = part + ((uint64_t)split * tiles + tile) * 4096u + wave * 2048u + lane;
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 8; y++) out[(x * 8 + y) * 64] = acc[x][y];
}

// C[i][j] += sum over the row slices of tile (ib, jb)'s partials. Thread = one accumulator fragment (4 rows x 1 column):
// fragment f = (wave, x, y, lane) holds D[row = 4 kg + jj][col = m] of sub-tile (x, y), values exact integers in float.
__global__ void __launch_bounds__(256) kin_reduce_kernel(const kin_f32x4* part, uint32_t tiles, uint32_t splits, uint32_t S_pad,
                                                         unsigned long long* C) {
    uint32_t tix = blockIdx.y, ib = 0;
    const uint32_t nt = S_pad / 128u;
    while (tix >= nt - ib) {
        tix -= nt - ib;
        ib++;
    }
    const uint32_t jb = ib + tix;
    const uint32_t f = blockIdx.x * 256u + threadIdx.x;  // 0 .. 4095
    unsigned long long v[4] = {0, 0, 0, 0};
    for (uint32_t sp = 0; sp < splits; sp++) {
        const kin_f32x4 t = part[((uint64_t)sp * tiles + blockIdx.y) * 4096u + f];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) v[jj] += (unsigned long long)t[jj];  // < 2^24 rows per slice: exact
    }
    const uint32_t wave = f >> 11, xy = (f >> 6) & 31u, lane = f & 63u;
    const uint32_t x = xy >> 3, y = xy & 7u, m = lane & 15u, kg = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const uint32_t i = ib * 128u + wave * 64u + x * 16u + kg * 4u + jj, j = jb * 128u + y * 16u + m;
        if (v[jj]) C[(uint64_t)i * S_pad + j] += v[jj];
    }
}

size_t kin_transpose_lds_bytes(uint64_t file_stride_w, uint32_t S_pad, uint32_t rpb) {
    return ((size_t)rpb * 2u * file_stride_w + (size_t)S_pad * (rpb / 32u) + 4u * rpb) * 4u;
}
[[maybe_unused]] static size_t kin_transpose_lds_bytes_direct(uint32_t S_pad, uint32_t rpb) { return ((size_t)S_pad * (rpb / 32u) + 4u * rpb) * 4u; }

// Rows per transpose block: the most (256, 128, 64) whose verbatim rows + planes fit the 160 KB of LDS; 0 = none does.
uint32_t kin_transpose_rows_per_block(uint64_t file_stride_w, uint32_t S_pad) {
    static const uint32_t rpb_env = (uint64_t)exp_int("KGWAS_KIN_RPB", 0u);  // experiments
    if (rpb_env && kin_transpose_lds_bytes(file_stride_w, S_pad, rpb_env) <= 160u * 1024u) return rpb_env;
    for (uint32_t rpb : {256u, 128u, 64u})
        if (kin_transpose_lds_bytes(file_stride_w, S_pad, rpb) <= 160u * 1024u) return rpb;
    return 0u;
}

hipError_t launch_kin_transpose(const uint64_t* file_rows, uint64_t file_stride_w, uint64_t n_rows, uint32_t S_f,
                                uint32_t S_pad, uint32_t min_count, uint32_t* T, uint64_t n_rw, unsigned long long* n_used,
                                hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    const uint32_t rpb = kin_transpose_rows_per_block(file_stride_w, S_pad);
    if (!rpb) return hipErrorInvalidValue;  // kgwas_kinship_create rejects such sessions with a message
    static const uint32_t tpr_env = (uint64_t)exp_int("KGWAS_KIN_TPR", 0u);  // experiments
    const uint32_t tpr = tpr_env ? tpr_env : 4u;  // threads per row (1: 3.55 ms per 8 M rows x 1135, 2: 3.16, 4: 3.11)
    // KGWAS_KIN_DIRECT=1 (experiments): rows straight into registers where a thread's part of a row is at most KIN_DIRECT_DW
    // dwords (up to 1536 accessions at four threads per row) instead of through the verbatim copy in LDS - four blocks per CU
    // instead of two, one barrier fewer: measured SLOWER, 3.30 against 3.13 ms per 8 M rows x 1135 (both kernels; the strided
    // row loads cost more than the LDS passes they replace)
#ifdef KGWAS_EXPERIMENTS
    static const bool direct_ok = exp_int("KGWAS_KIN_DIRECT", 0) != 0;
    const bool direct = direct_ok && (S_pad / 32u + tpr - 1u) / tpr <= KIN_DIRECT_DW;
    const size_t lds = direct ? kin_transpose_lds_bytes_direct(S_pad, rpb) : kin_transpose_lds_bytes(file_stride_w, S_pad, rpb);
    const void* fn = direct ? (const void*)kin_transpose_kernel<true> : (const void*)kin_transpose_kernel<false>;
#else  // (the shipped library holds the LDS-staged form only)
    const size_t lds = kin_transpose_lds_bytes(file_stride_w, S_pad, rpb);
    const void* fn = (const void*)kin_transpose_kernel<false>;
#endif
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // every word of T up to n_rw is written (blocks beyond the rows write zeros): n_rw / 8 tiles x 256 / rpb blocks
    const dim3 grid((uint32_t)(n_rw / 8u * (256u / rpb))), block(tpr * rpb);
#ifdef KGWAS_EXPERIMENTS
    if (direct)
        hipLaunchKernelGGL(kin_transpose_kernel<true>, grid, block, lds, st, file_rows, file_stride_w, n_rows, S_f, S_pad, min_count, T, n_rw, n_used, rpb);
    else
#endif
        hipLaunchKernelGGL(kin_transpose_kernel<false>, grid, block, lds, st, file_rows, file_stride_w, n_rows, S_f, S_pad, min_count, T, n_rw, n_used, rpb);
    return hipGetLastError();
}

// Row slices per 128 x 128 tile and plane words per slice for a chunk of n_rw plane words.
static void kin_gram_split(uint64_t n_rw, uint32_t tiles, uint64_t* per_out, uint32_t* splits_out) {
    static const uint64_t blocks_env = (uint64_t)exp_int("KGWAS_KIN_BLOCKS", 0);  // experiments
    // Row slices per tile: a CU holds four of these 2-wave blocks (two waves per SIMD), the chip 1024, and a launch
    // that is not close to a whole number of such rounds leaves CUs idle at its end (3105 blocks measured 5.5 ms per
    // 8 M rows x 1135 samples, 2025 blocks 5.0): as many slices as make about two full rounds.
    uint64_t want = std::max<uint64_t>(1, (blocks_env ? blocks_env : 1024ull) / tiles);
    uint64_t per = (n_rw + want - 1) / want;
    per = ((per + KIN_KC - 1) / KIN_KC) * KIN_KC;
    if (per < KIN_KC * 4) per = KIN_KC * 4;
    *per_out = per;
    *splits_out = (uint32_t)((n_rw + per - 1) / per);
}

// one 64 KB partial per (tile, row slice); a chunk never has more slices than kin_gram_split's `want`
size_t kin_gram_scratch_bytes(uint32_t S_pad) {
    const uint32_t nt = S_pad / 128u;
    const uint32_t tiles = nt * (nt + 1u) / 2u;
    uint64_t per;
    uint32_t splits;
    kin_gram_split(1ull << 40, tiles, &per, &splits);  // (n_rw >> want: splits == want)
    return (size_t)splits * tiles * 65536u;
}

hipError_t launch_kin_gram(const uint32_t* T, uint64_t n_rw, uint32_t S_pad, unsigned long long* C, void* scratch, size_t scratch_bytes,
                           hipStream_t st) {
    if (n_rw == 0) return hipSuccess;
    const uint32_t nt = S_pad / 128u;
    const uint32_t tiles = nt * (nt + 1u) / 2u;
    uint64_t per;
    uint32_t splits;
    kin_gram_split(n_rw, tiles, &per, &splits);
    if ((size_t)splits * tiles * 65536u > scratch_bytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kin_gram_kernel, dim3(tiles * splits), dim3(128), 0, st, T, n_rw, S_pad, (kin_f32x4*)scratch, per, tiles);
    hipLaunchKernelGGL(kin_reduce_kernel, dim3(16, tiles), dim3(256), 0, st, (const kin_f32x4*)scratch, tiles, splits, S_pad, C);
    return hipGetLastError();
}

}  // namespace kgwas
