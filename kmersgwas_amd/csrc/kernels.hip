// kernels.hip — hand-written gfx950 (CDNA4 / MI355X) kernels of libkgwas.
//
// What they compute is fixed by the reference (paths relative to the reference tree):
//   squeeze      : MultipleKmersDataBases::load_kmers' per-bit column gather
//                  (src/kmers_multiple_databases.cpp:125-132)
//   score_*      : calculate_unsqueezed_popcnt + the MAC predicate (:119, :149-154) and
//                  calculate_kmer_score (:327-363) for every (k-mer, phenotype column)
//   kin_*        : update_emma_kinshhip_calculation (:418-438)
// How they compute it is MI355X-first:
//   * The reference's float32 accumulation order is four independent sequential chains per
//     (k-mer, phenotype): SSE lane l walks samples 128b+32l+31-s for b = 0.., s = 0..31.
//     Adding (bit ? y : +0.0f) is bit-identical to fmaf((float)bit, y, acc), and gfx950's
//     v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain, so the exact reference
//     scores come straight out of the matrix cores: A = bits of 16 k-mers (one f32 0/1 per
//     lane), B = 16 phenotype columns of y in chain order (from LDS), one accumulator per
//     SSE lane. No re-scoring pass is needed and no tolerance is involved.
//   * The phenotype tile (16 columns x L floats, chain-step major) lives in LDS for the
//     whole block; table rows stream from HBM in their on-disk layout (direct mode) and are
//     never staged or re-tiled.
//   * blockIdx -> (row block, column tile) is XCD-aware: the column tiles that share a row
//     block run on the same XCD back to back, so the table is read from HBM once and
//     re-read from that XCD's L2.
//   * Scores are finished in double precision exactly as the reference does (no FMA
//     contraction: explicit __dmul_rn/__dsub_rn and -ffp-contract=off).
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off

#include "kernels.h"
#include "synth.h"

namespace kgwas {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// Shared epilogue: the double-precision tail of calculate_kmer_score (:359-361) and the
// hand-off to the host-side BestAssociationsHeap replay.
//   dense mode : every score is written (0 for rows the MAC filter drops).
//   sparse mode: a record is shipped only if score > thr[p], thr[p] being a stale value of
//                the heap minimum (BestAssociationsHeap::add_association is a no-op for
//                everything else once the heap is full, src/best_associations_heap.cpp:49-58).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void finish_pair(const ScoreArgs& a, uint64_t r, uint32_t p, float yf, uint32_t n1,
                                            bool pass) {
    const double N = (double)a.S;
    const double N1 = (double)n1;
    const double yigi = (double)yf;
    const double rr = __dsub_rn(__dmul_rn(N, yigi), __dmul_rn(N1, (double)a.sums[p]));
    const double q = __dmul_rn(rr, rr);
    const double d = __dsub_rn(__dmul_rn(N, N1), __dmul_rn(N1, N1));  // exact integers
    if (a.dense) {
        a.dense[(uint64_t)p * a.n_rows + r] = pass ? (q / d) : 0.0;
        return;
    }
    if (!pass) return;
    const double t = a.thr[p];
    // Conservative prefilter without the division: fl(q/d) > t implies q > t*d >= lim.
    double lim = __dmul_rn(t, d);
    lim = __dsub_rn(lim, __dmul_rn(fabs(lim), 0x1p-40));
    if (q >= lim) {
        const double s = q / d;  // correctly rounded IEEE division, as divsd on the host
        if (s > t) {
            const uint32_t slot = atomicAdd(&a.cand_cnt[p], 1u);
            if (slot < a.cap) {
                Cand c;
                c.kmer = a.file_rows[r * a.file_stride_w];
                c.score = s;
                c.row = a.first_row + r;
                a.cand[(uint64_t)p * a.cap + slot] = c;
            }
        }
    }
}

__device__ __forceinline__ bool mac_pass(const ScoreArgs& a, uint32_t n1) {
    // :119  (popcnt >= mac) && (popcnt <= accessions - mac), size_t arithmetic
    return (a.S >= a.min_count) && (n1 >= a.min_count) && (n1 <= a.S - a.min_count);
}

// ------------------------------------------------------------------------------------------
// Exact-order scorer on the vector ALU. One lane per k-mer row, PC phenotype columns per
// thread; y is wave-uniform and comes through scalar loads. 1 + 2*PC lane-ops per table bit.
// Used for few phenotype columns and for columns holding non-finite values (where
// 0*inf would poison the multiplicative MFMA formulation).
// ------------------------------------------------------------------------------------------
template <int PC>
__global__ void __launch_bounds__(256) score_valu_kernel(ScoreArgs a) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t ndw = 2u * a.W_m;
    const uint32_t ldw = ndw + 1u;  // odd stride: lane == row reads are bank-conflict free
    const uint32_t TR = blockDim.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * TR;

    for (uint32_t e = threadIdx.x; e < TR * ndw; e += TR) {
        const uint32_t rr = e / ndw, dw = e - rr * ndw;
        const uint64_t gr = row0 + rr;
        uint32_t v = 0;
        if (gr < a.n_rows && dw < a.src.avail_dw)
            v = a.src.base[gr * a.src.stride_dw + a.src.off_dw + dw] & a.dmask[dw];
        lds_u32[rr * ldw + dw] = v;
    }
    __syncthreads();

    const uint32_t rr = threadIdx.x;
    const uint64_t r = row0 + rr;
    const uint32_t* my = lds_u32 + rr * ldw;
    uint32_t n1 = 0;
    for (uint32_t dw = 0; dw < ndw; dw++) n1 += __popc(my[dw]);
    const bool in_range = r < a.n_rows;
    const bool pass = in_range && mac_pass(a, n1);

    const uint32_t p0 = blockIdx.y * PC;
    const uint32_t L = 64u * a.W_m;
    float acc[PC][4];
#pragma unroll
    for (int pc = 0; pc < PC; pc++)
#pragma unroll
        for (int l = 0; l < 4; l++) acc[pc][l] = 0.0f;

    const uint32_t nblk = a.W_m / 2u;
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t w[4];
#pragma unroll
        for (int l = 0; l < 4; l++) w[l] = my[4 * b + l];
        const float* yb = a.Yperm + (size_t)p0 * L + 128u * b;
#pragma unroll
        for (int s = 0; s < 32; s++) {
#pragma unroll
            for (int l = 0; l < 4; l++) {
                const int m = ((int)(w[l] << s)) >> 31;  // 0 or -1: bit 31-s of SSE lane l
#pragma unroll
                for (int pc = 0; pc < PC; pc++) {
                    const float y = yb[(size_t)pc * L + 4 * s + l];  // wave-uniform
                    acc[pc][l] = acc[pc][l] + __int_as_float(m & __float_as_int(y));
                }
            }
        }
    }

    if (blockIdx.y == 0 && in_range) {
        if (a.n1_out) a.n1_out[r] = n1;
        if (a.kmer_out) a.kmer_out[r] = a.file_rows[r * a.file_stride_w];
        if (pass && a.tested) atomicAdd(a.tested, 1ull);
    }
    if (!in_range) return;
#pragma unroll
    for (int pc = 0; pc < PC; pc++) {
        const uint32_t p = p0 + pc;
        if (p >= a.n_pheno) break;
        const float yf = ((acc[pc][0] + acc[pc][1]) + acc[pc][2]) + acc[pc][3];  // :358, float adds
        finish_pair(a, r, p, yf, n1, pass);
    }
}

// ------------------------------------------------------------------------------------------
// Exact-order scorer on the matrix cores (v_mfma_f32_16x16x4_f32).
//   block  = 4 waves sharing one 16-column phenotype tile in LDS
//   wave   = 2 row tiles of 16 k-mers per pass -> 8 independent accumulators
//            (2 row tiles x 4 SSE lanes), well past the 40-cycle dependent latency
//   lane   = (m = lane & 15 : k-mer row of the tile, kk = lane >> 4 : k index 0..3)
//   A[m][kk] = bit 31-(4t+kk) of the SSE-lane sub-word, as 0.0f / 1.0f
//   B[kk][n] = y_n[128b + 32l + 31 - (4t+kk)]           (LDS, chain-step major)
//   D reg j  = row (lane>>4)*4 + j, column lane & 15
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) score_mfma_kernel(ScoreArgs a, uint32_t rows_per_block, uint32_t n_rowblocks,
                                                         uint32_t n_ctiles) {
    extern __shared__ float ylds[];  // [L][16]
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, q = bid >> 3;
    const uint32_t ct = q % n_ctiles;
    const uint32_t rb = (q / n_ctiles) * 8u + xcd;
    if (rb >= n_rowblocks) return;  // whole block leaves before any barrier

    const uint32_t L = 64u * a.W_m;
    {
        const float4* src = reinterpret_cast<const float4*>(a.Ymfma + (size_t)ct * L * 16u);
        float4* dst = reinterpret_cast<float4*>(ylds);
        for (uint32_t i = threadIdx.x; i < L * 4u; i += 256u) dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t kk = lane >> 4;
    const uint32_t m = lane & 15u;
    const uint32_t nblk = a.W_m / 2u;
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;

    for (uint32_t ps = 0; ps * 128u < rows_per_block; ps++) {
        const uint64_t rbase = blk_row0 + (uint64_t)ps * 128u + wave * 32u;
        if (rbase >= a.n_rows) break;  // wave-uniform

        const uint32_t* rp[2];
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            uint64_t r = rbase + rt * 16u + m;
            if (r >= a.n_rows) r = a.n_rows - 1;  // clamp loads; results discarded below
            rp[rt] = a.src.base + r * a.src.stride_dw + a.src.off_dw;
        }

        f32x4 acc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int l = 0; l < 4; l++) acc[rt][l] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t n1[2] = {0u, 0u};

        for (uint32_t b = 0; b < nblk; b++) {
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
                n1[rt] += __popc(w[rt][0]) + __popc(w[rt][1]) + __popc(w[rt][2]) + __popc(w[rt][3]);
#pragma unroll
                for (int l = 0; l < 4; l++) w[rt][l] <<= kk;  // lane's k index folded into the word
            }
            const float* yb = ylds + (size_t)b * 2048u + lane;  // ((b*4+l)*32 + 4t)*16 + lane
#pragma unroll
            for (int t = 0; t < 8; t++) {
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    const float Bv = yb[l * 512 + t * 64];
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) {
                        const float Av = (float)((w[rt][l] >> (31 - 4 * t)) & 1u);
                        acc[rt][l] = __builtin_amdgcn_mfma_f32_16x16x4f32(Av, Bv, acc[rt][l], 0, 0, 0);
                    }
                }
            }
        }

        // Row bookkeeping comes from the kk == 0 lanes (lane == row of the tile).
        if (ct == 0 && kk == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const uint64_t r = rbase + rt * 16u + m;
                if (r < a.n_rows) {
                    if (a.n1_out) a.n1_out[r] = n1[rt];
                    if (a.kmer_out) a.kmer_out[r] = a.file_rows[r * a.file_stride_w];
                    if (a.tested && mac_pass(a, n1[rt])) atomicAdd(a.tested, 1ull);
                }
            }
        }

        const uint32_t p = ct * 16u + m;
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t trow = kk * 4u + j;  // D row held in register j
                const uint32_t n1r = __shfl(n1[rt], trow);
                const uint64_t r = rbase + rt * 16u + trow;
                const float yf = ((acc[rt][0][j] + acc[rt][1][j]) + acc[rt][2][j]) + acc[rt][3][j];  // :358
                if (r < a.n_rows && p < a.n_pheno) finish_pair(a, r, p, yf, n1r, mac_pass(a, n1r));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Squeeze: gather the phenotyped columns of 64 rows into phenotype order. Rows go through
// LDS both ways so that HBM sees only coalesced traffic; colmap is wave-uniform.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) squeeze_kernel(const uint64_t* file_rows, uint64_t file_stride_w,
                                                      uint64_t n_rows, const uint32_t* colmap, uint32_t W_m,
                                                      uint32_t W_f, uint32_t* out) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t in_dw = 2u * W_f, in_ld = in_dw + 1u;
    const uint32_t out_dw = 2u * W_m, out_ld = out_dw + 1u;
    uint32_t* lin = lds_u32;
    uint32_t* lout = lds_u32 + 64u * in_ld;
    const uint64_t row0 = (uint64_t)blockIdx.x * 64u;
    const uint32_t* fr = reinterpret_cast<const uint32_t*>(file_rows);

    for (uint32_t e = threadIdx.x; e < 64u * in_dw; e += 256u) {
        const uint32_t rr = e / in_dw, dw = e - rr * in_dw;
        const uint64_t gr = row0 + rr;
        lin[rr * in_ld + dw] = (gr < n_rows) ? fr[gr * file_stride_w * 2u + 2u + dw] : 0u;
    }
    __syncthreads();
    const uint32_t rr = threadIdx.x & 63u;
    const uint32_t g = threadIdx.x >> 6;  // wave id: colmap reads are wave-uniform
    for (uint32_t d = g; d < out_dw; d += 4u) {
        uint32_t o = 0;
#pragma unroll 8
        for (uint32_t j = 0; j < 32u; j++) {
            const uint32_t c = colmap[32u * d + j];
            if (c != 0xFFFFFFFFu) o |= ((lin[rr * in_ld + (c >> 5)] >> (c & 31u)) & 1u) << j;
        }
        lout[rr * out_ld + d] = o;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < 64u * out_dw; e += 256u) {
        const uint32_t r2 = e / out_dw, dw = e - r2 * out_dw;
        const uint64_t gr = row0 + r2;
        if (gr < n_rows) out[gr * out_dw + dw] = lout[r2 * out_ld + dw];
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic rows: one thread per 64-bit word.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) synth_kernel(uint64_t* rows, uint64_t first_row, uint64_t n_rows,
                                                    uint64_t n_acc, uint64_t seed) {
    const uint64_t W = 1 + (n_acc + 63) / 64;
    const uint64_t total = n_rows * W;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / W;
        const uint32_t w = (uint32_t)(i - r * W);
        rows[i] = synth_word(seed, first_row + r, w, n_acc);
    }
}

// ------------------------------------------------------------------------------------------
// Kinship, step 1: MAC filter over all S_f columns + bit transpose.
// T[c][rw] (u32) holds sample c's presence bits for rows 32*rw .. 32*rw+31 of the launch;
// rows failing the filter contribute all-zero bits, i.e. nothing to any Hamming distance.
// A block covers 512 rows (two 256-row halves) so that each sample's output is one 64-byte line.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kin_transpose_kernel(const uint64_t* file_rows, uint64_t file_stride_w,
                                                            uint64_t n_rows, uint32_t S_f, uint32_t S_pad,
                                                            uint32_t min_count, uint32_t* T, uint64_t n_rw,
                                                            unsigned long long* n_used) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t W_f = (S_f + 63u) / 64u;
    const uint32_t in_dw = 2u * W_f, in_ld = in_dw + 1u;
    uint32_t* lin = lds_u32;                     // [256][in_ld]
    uint32_t* lout = lds_u32 + 256u * in_ld;     // [S_pad][16]
    const uint64_t blk_row0 = (uint64_t)blockIdx.x * 512u;
    const uint32_t* fr = reinterpret_cast<const uint32_t*>(file_rows);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

    for (uint32_t half = 0; half < 2u; half++) {
        const uint64_t row0 = blk_row0 + half * 256u;
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < 256u * in_dw; e += 256u) {
            const uint32_t rr = e / in_dw, dw = e - rr * in_dw;
            const uint64_t gr = row0 + rr;
            lin[rr * in_ld + dw] = (gr < n_rows) ? fr[gr * file_stride_w * 2u + 2u + dw] : 0u;
        }
        __syncthreads();
        const uint32_t rr = threadIdx.x;
        const uint64_t r = row0 + rr;
        uint32_t n1 = 0;
        for (uint32_t dw = 0; dw < in_dw; dw++) n1 += __popc(lin[rr * in_ld + dw]);
        // src/emma_kinship_kmers.cpp:83,89 -> load_kmers' predicate with all S_f columns
        const bool pass = (r < n_rows) && (S_f >= min_count) && (n1 >= min_count) && (n1 <= S_f - min_count);
        if (pass) atomicAdd(n_used, 1ull);
        for (uint32_t c = 0; c < S_pad; c++) {
            const bool bit = pass && (c < S_f) && ((lin[rr * in_ld + (c >> 5)] >> (c & 31u)) & 1u);
            const unsigned long long bal = __ballot(bit);
            if (lane == 0) {
                lout[c * 16u + half * 8u + wave * 2u + 0u] = (uint32_t)bal;
                lout[c * 16u + half * 8u + wave * 2u + 1u] = (uint32_t)(bal >> 32);
            }
        }
    }
    __syncthreads();
    const uint64_t rw0 = (uint64_t)blockIdx.x * 16u;
    for (uint32_t e = threadIdx.x; e < S_pad * 16u; e += 256u) {
        const uint32_t c = e >> 4, k = e & 15u;
        if (rw0 + k < n_rw) T[(uint64_t)c * n_rw + rw0 + k] = lout[e];
    }
}

// ------------------------------------------------------------------------------------------
// Kinship, step 2: H[i][j] += sum_rw popcount(T[i][rw] ^ T[j][rw]) — a "GEMM" whose
// multiply-add is xor + v_bcnt_u32_b32. 64x64 tile per block, 4x4 per thread, split over rw.
// 1 ^ g_i ^ g_j summed over the used rows is n_used - H[i][j].
// ------------------------------------------------------------------------------------------
#define KIN_KC 32u
__global__ void __launch_bounds__(256) kin_gram_kernel(const uint32_t* T, uint64_t n_rw, uint32_t S_pad,
                                                       unsigned long long* H, uint32_t n_t1d, uint64_t rw_per_split) {
    __shared__ uint32_t A[64][KIN_KC + 1];
    __shared__ uint32_t B[64][KIN_KC + 1];
    // decode lower-triangular tile index
    uint32_t tix = blockIdx.x, ib = 0;
    while (tix >= ib + 1u) {
        tix -= ib + 1u;
        ib++;
    }
    const uint32_t jb = tix;
    (void)n_t1d;
    const uint64_t k_begin = (uint64_t)blockIdx.y * rw_per_split;
    uint64_t k_end = k_begin + rw_per_split;
    if (k_end > n_rw) k_end = n_rw;
    const uint32_t ti = threadIdx.x >> 4, tj = threadIdx.x & 15u;
    uint32_t acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++) acc[x][y] = 0u;

    for (uint64_t k0 = k_begin; k0 < k_end; k0 += KIN_KC) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < 64u * KIN_KC; e += 256u) {
            const uint32_t row = e / KIN_KC, kw = e % KIN_KC;
            const bool ok = (k0 + kw) < k_end;
            A[row][kw] = ok ? T[(uint64_t)(ib * 64u + row) * n_rw + k0 + kw] : 0u;
            B[row][kw] = ok ? T[(uint64_t)(jb * 64u + row) * n_rw + k0 + kw] : 0u;
        }
        __syncthreads();
#pragma unroll 4
        for (uint32_t kw = 0; kw < KIN_KC; kw++) {
            uint32_t av[4], bv[4];
#pragma unroll
            for (int x = 0; x < 4; x++) av[x] = A[ti * 4 + x][kw];
#pragma unroll
            for (int y = 0; y < 4; y++) bv[y] = B[tj * 4 + y][kw];
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 4; y++) acc[x][y] += __popc(av[x] ^ bv[y]);
        }
    }
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t i = ib * 64u + ti * 4u + x, j = jb * 64u + tj * 4u + y;
            if (acc[x][y]) {
                atomicAdd(&H[(uint64_t)i * S_pad + j], (unsigned long long)acc[x][y]);
                if (ib != jb) atomicAdd(&H[(uint64_t)j * S_pad + i], (unsigned long long)acc[x][y]);
            }
        }
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------
static hipError_t ensure_dyn_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

hipError_t launch_score_valu(const ScoreArgs& a, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const uint32_t ldw = 2u * a.W_m + 1u;
    uint32_t TR = 256;
    while (TR > 64 && (size_t)TR * ldw * 4u > 150u * 1024u) TR >>= 1;
    const size_t lds = (size_t)TR * ldw * 4u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    const uint32_t gx = (uint32_t)((a.n_rows + TR - 1) / TR);
    hipError_t e;
    if (a.n_pheno == 1) {
        if ((e = ensure_dyn_lds((const void*)score_valu_kernel<1>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(score_valu_kernel<1>, dim3(gx, 1), dim3(TR), lds, st, a);
    } else {
        if ((e = ensure_dyn_lds((const void*)score_valu_kernel<4>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(score_valu_kernel<4>, dim3(gx, (a.n_pheno + 3) / 4), dim3(TR), lds, st, a);
    }
    return hipGetLastError();
}

size_t mfma_lds_bytes(uint32_t W_m) { return (size_t)64u * W_m * 16u * sizeof(float); }

hipError_t launch_score_mfma(const ScoreArgs& a, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const size_t lds = mfma_lds_bytes(a.W_m);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = ensure_dyn_lds((const void*)score_mfma_kernel, lds);
    if (e != hipSuccess) return e;
    const uint32_t n_ctiles = (a.n_pheno + 15u) / 16u;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
    const uint32_t groups = (n_rowblocks + 7u) / 8u;
    const uint32_t grid = groups * n_ctiles * 8u;
    hipLaunchKernelGGL(score_mfma_kernel, dim3(grid), dim3(256), lds, st, a, rows_per_block, n_rowblocks, n_ctiles);
    return hipGetLastError();
}

hipError_t launch_squeeze(const uint64_t* file_rows, uint64_t file_stride_w, uint64_t n_rows, const uint32_t* colmap,
                          uint32_t W_m, uint32_t W_f, uint32_t* out, hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    const size_t lds = (size_t)64u * ((2u * W_f + 1u) + (2u * W_m + 1u)) * 4u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = ensure_dyn_lds((const void*)squeeze_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(squeeze_kernel, dim3((uint32_t)((n_rows + 63) / 64)), dim3(256), lds, st, file_rows,
                       file_stride_w, n_rows, colmap, W_m, W_f, out);
    return hipGetLastError();
}

hipError_t launch_synth(uint64_t* rows, uint64_t first_row, uint64_t n_rows, uint64_t n_acc, uint64_t seed,
                        hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    const uint64_t total = n_rows * (1 + (n_acc + 63) / 64);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, rows, first_row, n_rows, n_acc, seed);
    return hipGetLastError();
}

hipError_t launch_kin_transpose(const uint64_t* file_rows, uint64_t file_stride_w, uint64_t n_rows, uint32_t S_f,
                                uint32_t S_pad, uint32_t min_count, uint32_t* T, uint64_t n_rw,
                                unsigned long long* n_used, hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    const uint32_t W_f = (S_f + 63u) / 64u;
    const size_t lds = ((size_t)256u * (2u * W_f + 1u) + (size_t)S_pad * 16u) * 4u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = ensure_dyn_lds((const void*)kin_transpose_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kin_transpose_kernel, dim3((uint32_t)((n_rows + 511) / 512)), dim3(256), lds, st, file_rows,
                       file_stride_w, n_rows, S_f, S_pad, min_count, T, n_rw, n_used);
    return hipGetLastError();
}

hipError_t launch_kin_gram(const uint32_t* T, uint64_t n_rw, uint32_t S_pad, unsigned long long* H, hipStream_t st) {
    if (n_rw == 0) return hipSuccess;
    const uint32_t nt = S_pad / 64u;
    const uint32_t tiles = nt * (nt + 1u) / 2u;
    // enough k-splits to fill 256 CUs several times over, each split a multiple of KIN_KC words
    uint64_t want = (256ull * 8ull + tiles - 1) / tiles;
    uint64_t per = (n_rw + want - 1) / want;
    per = ((per + KIN_KC - 1) / KIN_KC) * KIN_KC;
    if (per < KIN_KC * 4) per = KIN_KC * 4;
    const uint32_t splits = (uint32_t)((n_rw + per - 1) / per);
    hipLaunchKernelGGL(kin_gram_kernel, dim3(tiles, splits), dim3(256), 0, st, T, n_rw, S_pad, H, nt, per);
    return hipGetLastError();
}

}  // namespace kgwas
