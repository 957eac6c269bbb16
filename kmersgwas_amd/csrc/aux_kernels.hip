// aux_kernels.hip — the gfx950 kernels around the scorers: column squeeze, synthetic table
// generator and the two kinship kernels.
//   squeeze : MultipleKmersDataBases::load_kmers' per-bit column gather
//             (src/kmers_multiple_databases.cpp:125-132), only needed when the phenotyped
//             accessions are not the table's leading columns in order
//   kin_*   : update_emma_kinshhip_calculation (src/kmers_multiple_databases.cpp:418-438)
#include "score_common.h"
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
    thr_update_block(hist, hist_base, bins, topn, thr_host, thr, blockIdx.x);
}

// One digit of the radix select, by wave 0: the highest bin b with (counts of the bins above b) < need <= (those + hist[b]);
// need becomes what is still wanted inside b, the prefix takes b as its next byte. (One thread walking the 256 bins from the top
// was a chain of dependent LDS reads - 9 us per digit, 72 of the 107 us the selection took at the head of every scan.) Lane l holds
// bins 4 l .. 4 l + 3; a suffix sum over the lanes finds the lane, the lane finds the bin. The walk of the first version ended at
// bin 0 whatever was left: so does this one.
__device__ __forceinline__ void pick_bin_from_top(const uint32_t* hist, unsigned long long prefix, unsigned long long* need_s,
                                                  unsigned long long* prefix_s) {
    const uint32_t l = threadIdx.x & 63u;
    const uint4 c = reinterpret_cast<const uint4*>(hist)[l];
    const unsigned long long mine = (unsigned long long)c.x + c.y + c.z + c.w;
    unsigned long long suf = mine;  // lanes l .. 63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long x = __shfl_down(suf, d);
        if (l + (uint32_t)d < 64u) suf += x;
    }
    const unsigned long long need = *need_s;
    const unsigned long long above = suf - mine;
    const bool here = above < need && need <= suf;
    const unsigned long long m = __ballot(here);
    if (m ? here : l == 0u) {
        unsigned long long left = need - above;  // (m == 0: fewer than `need` keys in all - cannot happen; bin 0, as the walk did)
        uint32_t j = 3;  // (no array indexed by j: that would live in scratch memory, 0.2 ms per launch)
        if (c.w < left) {
            left -= c.w;
            j = 2;
            if (c.z < left) {
                left -= c.z;
                j = 1;
                if (c.y < left) {
                    left -= c.y;
                    j = 0;
                }
            }
        }
        *need_s = left;
        *prefix_s = (prefix << 8) | (unsigned long long)(4u * l + j);
    }
}

// ------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------
// ---- dense start: the heaps' minima-to-be, selected on the device ----------------------------------------------------
// After the first dense chunk every MAC-passing row is pushed into every heap (src/best_associations_heap.cpp:43-59), so
// a column's heap minimum afterwards is simply the topn[p]-th largest of the chunk's scores - if the chunk has that many
// MAC-passing rows. One block per column finds it by radix select over the doubles' bit patterns (order-preserving key),
// so the sparse chunks can start against these thresholds while the host is still pushing the dense rows.
// info[0] = MAC-passing rows of the chunk, info[1] = 1 if any of their scores is NaN (the caller then takes the plain path).
__global__ void __launch_bounds__(256) dense_select_kernel(const double* dense, const uint32_t* n1, uint32_t n_rows, uint32_t S,
                                                           uint32_t min_count, const uint64_t* topn, double* thr_a, double* thr_b,
                                                           double* thr_host_copy, uint32_t* info) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[256];
    __shared__ unsigned long long prefix_s, need_s;
    __shared__ uint32_t kept_s, nan_s;
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const double* sc = dense + (size_t)p * n_rows;
    auto pass_mac = [&](uint32_t r) {
        const uint32_t c = n1[r];
        return S >= min_count && c >= min_count && c <= S - min_count;
    };
    auto key_of = [](double v) {
        unsigned long long b = (unsigned long long)__double_as_longlong(v);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    };
    if (t == 0) {
        kept_s = 0;
        nan_s = 0;
    }
    __syncthreads();
    {
        uint32_t k = 0, bad = 0;
        for (uint32_t r = t; r < n_rows; r += 256u)
            if (pass_mac(r)) {
                k++;
                const double v = sc[r];
                bad |= (v != v) ? 1u : 0u;
            }
        atomicAdd(&kept_s, k);
        if (bad) atomicOr(&nan_s, 1u);
    }
    __syncthreads();
    const uint32_t kept = kept_s;
    if (p == 0 && t == 0) info[0] = kept;
    if (t == 0 && nan_s) atomicOr(&info[1], 1u);
    const unsigned long long N = topn[p];
    if (kept < N || nan_s) return;  // (block-uniform) not enough rows to fill this heap: the caller falls back
    if (t == 0) {
        prefix_s = 0;
        need_s = N;
    }
#pragma unroll 1
    for (int byte = 7; byte >= 0; byte--) {
        hist[t] = 0;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
        for (uint32_t r = t; r < n_rows; r += 256u)
            if (pass_mac(r)) {
                const unsigned long long k = key_of(sc[r]);
                if (byte == 7 || (k >> (8 * (byte + 1))) == prefix) atomicAdd(&hist[(uint32_t)(k >> (8 * byte)) & 255u], 1u);
            }
        __syncthreads();
        if (t < 64u) pick_bin_from_top(hist, prefix, &need_s, &prefix_s);
        __syncthreads();
    }
    if (t == 0) {
        const unsigned long long k = prefix_s;
        const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        const double v = __longlong_as_double((long long)b);
        thr_a[p] = v;
        thr_b[p] = v;
        thr_host_copy[p] = v;
    }
}

// The same selection for chunks of up to 16 x 1024 rows (the first dense chunk is 11 776 at N = 10 001), which is every scan's
// first 0.2 ms with the kernel above - nine passes over the column in L2, and the scores of a column share their sign, exponent
// and leading mantissa bits, so in the first passes all 256 threads' atomics land in one or two bins: a block of 1024 threads
// keeps the keys in registers (read once), and a wave whose active lanes all fall into one bin adds their count with one
// atomic: the dense start of a one-column scan 0.49-0.51 -> 0.33-0.41 ms (KGWAS_DSEL_REG=0: the kernel above).
constexpr int DSEL_KPT = 16;
__global__ void __launch_bounds__(1024) dense_select_reg_kernel(const double* dense, const uint32_t* n1, uint32_t n_rows, uint32_t S,
                                                                uint32_t min_count, const uint64_t* topn, double* thr_a, double* thr_b,
                                                                double* thr_host_copy, uint32_t* info) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[256];
    __shared__ unsigned long long prefix_s, need_s;
    __shared__ uint32_t kept_s, nan_s;
    const uint32_t p = blockIdx.x, t = threadIdx.x, lane = t & 63u;
    const double* sc = dense + (size_t)p * n_rows;
    if (t == 0) {
        kept_s = 0;
        nan_s = 0;
    }
    __syncthreads();
    unsigned long long key[DSEL_KPT];
    uint32_t valid = 0, kcnt = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < DSEL_KPT; i++) {
        const uint32_t r = t + (uint32_t)i * 1024u;
        key[i] = 0;
        if (r < n_rows) {
            const uint32_t c = n1[r];
            if (S >= min_count && c >= min_count && c <= S - min_count) {
                const double v = sc[r];
                bad |= (v != v) ? 1u : 0u;
                const unsigned long long b = (unsigned long long)__double_as_longlong(v);
                key[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
                valid |= 1u << i;
                kcnt++;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) kcnt += __shfl_xor(kcnt, o);
    if (lane == 0 && kcnt) atomicAdd(&kept_s, kcnt);
    if (__ballot(bad != 0u) && lane == 0) atomicOr(&nan_s, 1u);
    __syncthreads();
    const uint32_t kept = kept_s;
    if (p == 0 && t == 0) info[0] = kept;
    if (t == 0 && nan_s) atomicOr(&info[1], 1u);
    const unsigned long long N = topn[p];
    if (kept < N || nan_s) return;  // (block-uniform) not enough rows to fill this heap: the caller falls back
    if (t == 0) {
        prefix_s = 0;
        need_s = N;
    }
#pragma unroll 1
    for (int byte = 7; byte >= 0; byte--) {
        if (t < 256u) hist[t] = 0;
        __syncthreads();
        const unsigned long long prefix = prefix_s;
#pragma unroll
        for (int i = 0; i < DSEL_KPT; i++) {
            const bool act = ((valid >> i) & 1u) && (byte == 7 || (key[i] >> (8 * (byte + 1))) == prefix);
            const unsigned long long m = __ballot(act);
            if (!m) continue;  // (wave-uniform)
            const uint32_t bin = (uint32_t)(key[i] >> (8 * byte)) & 255u;
            const int first = __ffsll((long long)m) - 1;
            const uint32_t b0 = (uint32_t)__shfl((int)bin, first);
            if (__ballot(act && bin == b0) == m) {
                if ((int)lane == first) atomicAdd(&hist[b0], (uint32_t)__popcll(m));
            } else if (act) {
                atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        if (t < 64u) pick_bin_from_top(hist, prefix, &need_s, &prefix_s);
        __syncthreads();
    }
    if (t == 0) {
        const unsigned long long k = prefix_s;
        const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        const double v = __longlong_as_double((long long)b);
        thr_a[p] = v;
        thr_b[p] = v;
        thr_host_copy[p] = v;
    }
}

hipError_t launch_dense_select(const double* dense, const uint32_t* n1, uint32_t n_rows, uint32_t n_pheno, uint32_t S, uint32_t min_count,
                               const uint64_t* topn, double* thr_a, double* thr_b, double* thr_host_copy, uint32_t* info, hipStream_t st) {
    hipError_t e = hipMemsetAsync(info, 0, 2 * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    static const bool reg_ok = !(exp_int("KGWAS_DSEL_REG", 1) == 0);  // experiments: 0 = the first version
    if (reg_ok && n_rows <= (uint32_t)DSEL_KPT * 1024u)
        hipLaunchKernelGGL(dense_select_reg_kernel, dim3(n_pheno), dim3(1024), 0, st, dense, n1, n_rows, S, min_count, topn, thr_a, thr_b,
                           thr_host_copy, info);
    else
        hipLaunchKernelGGL(dense_select_kernel, dim3(n_pheno), dim3(256), 0, st, dense, n1, n_rows, S, min_count, topn, thr_a, thr_b,
                           thr_host_copy, info);
    return hipGetLastError();
}

// A chunk's records (three arrays) from their slot in HBM to their place in the pinned, mapped record ring, written by
// the GPU itself in ONE launch (three hipMemcpyAsync calls: three blits or SDMA transfers, and beside a streamed feed's
// 128 MiB host -> device pieces their completion came 2 ms late, scan_gpu.cpp fetch_records).
__global__ void __launch_bounds__(256) records_to_host_kernel(const unsigned long long* __restrict__ sc, const unsigned long long* __restrict__ km,
                                                              const uint32_t* __restrict__ rw, uint32_t n, unsigned long long* __restrict__ h_sc,
                                                              unsigned long long* __restrict__ h_km, uint32_t* __restrict__ h_rw) {
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        h_sc[i] = sc[i];
        h_km[i] = km[i];
        h_rw[i] = rw[i];
    }
}

// What the host needs behind a chunk before it orders the record copies - the chunk's counts (meta), the tested-row shards and
// the thresholds as they stand - written into the mapped host buffers by ONE launch instead of two or three hipMemcpyAsync
// calls (each a blit launch of its own on the scan's stream, 24 chunks per 100 M-row pass).
__global__ void __launch_bounds__(256) chunk_tail_kernel(const uint32_t* __restrict__ meta, uint32_t n_meta, uint32_t* __restrict__ h_meta,
                                                         const unsigned long long* __restrict__ tested, uint32_t n_tested,
                                                         unsigned long long* __restrict__ h_tested, const double* __restrict__ thr, uint32_t n_thr,
                                                         double* __restrict__ h_thr) {
    for (uint32_t i = threadIdx.x; i < n_meta; i += 256u) h_meta[i] = meta[i];
    for (uint32_t i = threadIdx.x; i < n_tested; i += 256u) h_tested[i] = tested[i];
    for (uint32_t i = threadIdx.x; i < n_thr; i += 256u) h_thr[i] = thr[i];
}

hipError_t launch_chunk_tail(const uint32_t* meta, uint32_t n_meta, uint32_t* h_meta, const unsigned long long* tested, uint32_t n_tested,
                             unsigned long long* h_tested, const double* thr, uint32_t n_thr, double* h_thr, hipStream_t st) {
    launch_last(chunk_tail_kernel, dim3(1), dim3(256), 0, st, meta, n_meta, h_meta, tested, n_tested, h_tested, thr, n_thr, h_thr);
    return hipGetLastError();
}

// The wide path's chunk end in ONE launch: block p raises thr[p] (thr_update_kernel's body) and writes it to the mapped buffer,
// block 0 also copies the counts and the tested-row shards (chunk_tail_kernel's work): the launch after the compaction is the
// chunk's last. The thresholds and what the host reads are what the two launches produced.
__global__ void __launch_bounds__(256) thr_tail_kernel(const uint32_t* hist, const uint32_t* hist_base, uint32_t bins, const uint64_t* topn,
                                                       const double* thr_host, double* thr, const uint32_t* __restrict__ meta, uint32_t n_meta,
                                                       uint32_t* __restrict__ h_meta, const unsigned long long* __restrict__ tested, uint32_t n_tested,
                                                       unsigned long long* __restrict__ h_tested, double* __restrict__ h_thr) {
    thr_update_block(hist, hist_base, bins, topn, thr_host, thr, blockIdx.x);  // (ends with a block barrier)
    if (h_thr && threadIdx.x == 0) h_thr[blockIdx.x] = thr[blockIdx.x];
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < n_meta; i += 256u) h_meta[i] = meta[i];
        for (uint32_t i = threadIdx.x; i < n_tested; i += 256u) h_tested[i] = tested[i];
    }
}

hipError_t launch_thr_tail(const uint32_t* hist, const uint32_t* hist_base, uint32_t bins, const uint64_t* topn, const double* thr_host, double* thr,
                           uint32_t n_pheno, const uint32_t* meta, uint32_t n_meta, uint32_t* h_meta, const unsigned long long* tested,
                           uint32_t n_tested, unsigned long long* h_tested, double* h_thr, hipStream_t st) {
    if (n_pheno == 0 || bins % 256u) return hipErrorInvalidValue;
    launch_last(thr_tail_kernel, dim3(n_pheno), dim3(256), 0, st, hist, hist_base, bins, topn, thr_host, thr, meta, n_meta, h_meta, tested,
                n_tested, h_tested, h_thr);
    return hipGetLastError();
}

hipError_t launch_records_to_host(const double* sc, const uint64_t* km, const uint32_t* rw, uint32_t n, double* h_sc, uint64_t* h_km, uint32_t* h_rw,
                                  hipStream_t st) {
    if (!n) return hipSuccess;
    static const uint32_t max_blocks = (uint64_t)exp_int("KGWAS_RECORD_BLOCKS", 64u);  // experiments (8 to 512 blocks: the same rates; few blocks leave the CUs to the filter)
    const uint32_t blocks = std::min<uint32_t>((n + 255u) / 256u, std::max(1u, max_blocks));
    hipLaunchKernelGGL(records_to_host_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const unsigned long long*>(sc),
                       reinterpret_cast<const unsigned long long*>(km), rw, n, reinterpret_cast<unsigned long long*>(h_sc),
                       reinterpret_cast<unsigned long long*>(h_km), h_rw);
    return hipGetLastError();
}

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

// Plain device -> mapped-host copy by the GPU itself (bytes a multiple of 4; both 16-byte aligned). The first SDMA
// device -> host transfer of a process blocked its hipMemcpyAsync call for 58 ms (the dense chunk's 9.5 MB of scores in
// a fresh `associate_kmers`), and later ones complete late beside host -> device pieces (launch_records_to_host).
__global__ void __launch_bounds__(256) copy_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, uint32_t tail_words) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += step) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail_words)
        reinterpret_cast<uint32_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint32_t*>(src + n16)[threadIdx.x];
}

hipError_t launch_copy_to_host(const void* src, void* dst_dev, size_t bytes, hipStream_t st) {
    if (!bytes) return hipSuccess;
    if (bytes % 4 || ((uintptr_t)src | (uintptr_t)dst_dev) % 16) return hipErrorInvalidValue;
    const size_t n16 = bytes / 16;
    const uint32_t blocks = (uint32_t)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 256);
    hipLaunchKernelGGL(copy_to_host_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst_dev), n16,
                       (uint32_t)((bytes % 16) / 4));
    return hipGetLastError();
}

}  // namespace kgwas
