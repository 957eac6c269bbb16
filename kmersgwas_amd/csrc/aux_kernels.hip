// aux_kernels.hip — the gfx950 kernels around the scorers: column squeeze, synthetic table
// generator and the two kinship kernels.
//   squeeze : MultipleKmersDataBases::load_kmers' per-bit column gather
//             (src/kmers_multiple_databases.cpp:125-132), only needed when the phenotyped
//             accessions are not the table's leading columns in order
//   kin_*   : update_emma_kinshhip_calculation (src/kmers_multiple_databases.cpp:418-438)
#include "kernels.h"
#include "synth.h"

namespace kgwas {

static hipError_t ensure_dyn_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
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
// Device-side threshold tracking. BestAssociationsHeap::add_association only changes a full heap
// when score > lowest_score (src/best_associations_heap.cpp:49-58); lowest_score after some rows is
// the N-th largest score seen, so ANY value v with at least N seen scores >= v is a lower bound of it.
// The scorers count every shipped candidate's score in a per-column histogram of its top bits;
// this kernel (one block per column, run between chunks) picks the highest bin boundary with >= N
// counted scores at or above it and raises thr[p] to it — no host round trip, so the GPU never waits
// for the replay. thr_host[p] (the exact minimum of the host heap as far as it has replayed) is
// folded in whenever it is higher. A NaN minimum (NaN inside a full heap) freezes the column:
// `score > NaN` is false on both sides.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) thr_update_kernel(const uint32_t* hist, const uint32_t* hist_base,
                                                         uint32_t bins, const uint64_t* topn, const double* thr_host,
                                                         double* thr) {
    __shared__ unsigned long long part[256];
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t per = bins / 256u;
    const uint32_t* h = hist + (uint64_t)p * bins;
    unsigned long long s = 0;
    for (uint32_t i = 0; i < per; i++) s += h[t * per + i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        const unsigned long long N = topn[p];
        unsigned long long run = 0;
        int seg = -1;
        for (int u = 255; u >= 0; u--) {
            if (run + part[u] >= N) {
                seg = u;
                break;
            }
            run += part[u];
        }
        double cur = thr[p];
        const double th = thr_host[p];
        if (seg >= 0) {
            uint32_t b = seg * per + per - 1;
            for (;; b--) {  // run = counted scores above this segment; walk down inside it
                run += h[b];
                if (run >= N || b == (uint32_t)seg * per) break;
            }
            const double v = __longlong_as_double((long long)((unsigned long long)(hist_base[p] + b) << HIST_SHIFT));
            if (v > cur) cur = v;
        }
        if (th != th || cur != cur)
            cur = __longlong_as_double(0x7FF8000000000000LL);
        else if (th > cur)
            cur = th;
        thr[p] = cur;
    }
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_thr_update(const uint32_t* hist, const uint32_t* hist_base, uint32_t bins, const uint64_t* topn,
                             const double* thr_host, double* thr, uint32_t n_pheno, hipStream_t st) {
    if (n_pheno == 0 || bins % 256u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(thr_update_kernel, dim3(n_pheno), dim3(256), 0, st, hist, hist_base, bins, topn, thr_host, thr);
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
