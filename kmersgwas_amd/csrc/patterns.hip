// patterns.hip — `--pattern_counter`: the number of distinct presence/absence patterns among the
// MAC-passing rows (src/associate_kmers.cpp:113-115,130-132,197-201).
//
// The reference inserts hash_presence_absence_pattern(row) of every kept row into a dense_hash_set
// and reports the set's size (src/kmers_multiple_databases.cpp:367-380): the hash is a boost-style
// combine of Hash64 (src/kmer_general.h:31-40) over the squeezed, zero-padded W_m words. Here one
// kernel hashes the rows of a chunk (a block's rows copied into LDS as they lie, one lane per row) and appends the hashes
// of the rows that pass the MAC predicate; at the end the hashes are radix-sorted on the device
// (hipCUB) and the distinct ones counted. Exact: it is the same 64-bit hash, and the count of
// distinct 64-bit values does not depend on how it is obtained.
#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace kgwas {

__device__ __forceinline__ uint64_t hash64(uint64_t key) {
    key ^= key >> 33;
    key *= 0xff51afd7ed558ccdULL;
    key ^= key >> 33;
    key *= 0xc4ceb9fe1a85ec53ULL;
    key ^= key >> 33;
    return key;
}

// A block takes PAT_TILES tiles of TR rows. A tile's rows are one contiguous piece of the source (rows of stride_dw / 2
// words): it is copied into LDS as it lies, 8 bytes per lane and turn (rows are 8-byte aligned), then lane t hashes row t out
// of LDS (a row's words are a row stride apart from the next lane's: 17 or 18 words at 1024 / 1135 samples, conflict-free
// or two-way). Words the source does not hold (beyond avail_dw) and bits beyond the last sample are masked away. The kept
// rows' hashes are collected in LDS and the block claims its slots in the output with ONE atomic on the launch's counter:
// device-scope atomics on one address serialise at the memory side at ~11 ns each - a row's own atomicAdd (93 M per 100 M
// rows), then one per wave (1.5 M), each cost the same 19 ms per 100 M rows, whatever else the kernel did (no hashing, no
// loads, no output: 24.0 / 23.7 / 24.2 ms per one-column pass against 23.8). The order of the hashes is immaterial (they are
// sorted for the distinct count).
constexpr uint32_t PAT_TILES = 2;  // (16: 11.4 ms per 100 M rows, 8: 8.8, 4: 6.0, 2: 4.8 - a block's tiles run one after the other)

__global__ void __launch_bounds__(256) pattern_hash_kernel(RowSrc src, const uint32_t* dmask, uint64_t n_rows, uint32_t S,
                                                           uint32_t W_m, uint32_t min_count, uint64_t* out,
                                                           unsigned long long* out_count) {
    extern __shared__ unsigned long long lds_u64[];  // [TR rows][stride_w] staged rows, then [PAT_TILES * TR] kept hashes
    __shared__ uint32_t n_kept;
    __shared__ unsigned long long out_base;
    const uint32_t TR = blockDim.x;
    const uint32_t stride_w = (uint32_t)(src.stride_dw >> 1), off_w = src.off_dw >> 1;
    unsigned long long* kept = lds_u64 + (size_t)TR * stride_w;
    if (threadIdx.x == 0) n_kept = 0;
    for (uint32_t tile = 0; tile < PAT_TILES; tile++) {
        const uint64_t row0 = ((uint64_t)blockIdx.x * PAT_TILES + tile) * TR;
        if (row0 >= n_rows) break;  // (uniform)
        const uint32_t rows_here = (uint32_t)(n_rows - row0 < TR ? n_rows - row0 : TR);
        __syncthreads();  // the stage is free (the tile before has been hashed); n_kept is initialised
        {
            const unsigned long long* g = reinterpret_cast<const unsigned long long*>(src.base) + row0 * stride_w;
            const uint32_t n_w = rows_here * stride_w;
            for (uint32_t e = threadIdx.x; e < n_w; e += TR) lds_u64[e] = __builtin_nontemporal_load(g + e);
        }
        __syncthreads();
        const bool live = threadIdx.x < rows_here;
        uint32_t n1 = 0;
        uint64_t seed = 0;
        if (live) {
            const unsigned long long* my = lds_u64 + threadIdx.x * stride_w + off_w;
            for (uint32_t w = 0; w < W_m; w++) {
                // (uniform per w: scalar loads)
                const uint64_t mlo = 2u * w < src.avail_dw ? dmask[2u * w] : 0u, mhi = 2u * w + 1u < src.avail_dw ? dmask[2u * w + 1u] : 0u;
                const uint64_t mask = mlo | (mhi << 32);
                const uint64_t word = mask ? (uint64_t)my[w] & mask : 0ull;  // (mask == 0: the word may lie beyond the staged rows)
                n1 += __popcll(word);
                seed ^= hash64(word) + 0x9e3779b97f4a7c15ULL + (seed << 6) + (seed >> 2);
            }
        }
        const bool keep = live && S >= min_count && n1 >= min_count && n1 <= S - min_count;
        const unsigned long long m = __ballot(keep);
        if (m) {
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t base = 0;
            if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(&n_kept, (uint32_t)__popcll(m));  // (LDS)
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (keep) kept[base + __popcll(m & ((1ull << lane) - 1ull))] = seed;
        }
    }
    __syncthreads();
    const uint32_t n = n_kept;
    if (threadIdx.x == 0 && n) out_base = atomicAdd(out_count, (unsigned long long)n);
    __syncthreads();
    if (n) {
        const unsigned long long ob = out_base;
        for (uint32_t i = threadIdx.x; i < n; i += TR) out[ob + i] = kept[i];
    }
}

__global__ void __launch_bounds__(256) count_distinct_sorted_kernel(const uint64_t* keys, uint64_t n,
                                                                   unsigned long long* distinct) {
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        local += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(distinct, local);
}

hipError_t launch_pattern_hash(const RowSrc& src, const uint32_t* dmask, uint64_t n_rows, uint32_t S, uint32_t W_m,
                               uint32_t min_count, uint64_t* out, unsigned long long* out_count, hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    if ((src.stride_dw & 1u) || (src.off_dw & 1u) || (reinterpret_cast<uintptr_t>(src.base) & 7u)) return hipErrorInvalidValue;  // rows of 64-bit words
    const size_t row_bytes = (size_t)src.stride_dw * 4u;
    uint32_t TR = 256;
    while (TR > 64 && (size_t)TR * (row_bytes + PAT_TILES * 8u) > 150u * 1024u) TR >>= 1;
    const size_t lds = (size_t)TR * (row_bytes + PAT_TILES * 8u);  // a tile's rows + the block's kept hashes
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)pattern_hash_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(pattern_hash_kernel, dim3((uint32_t)((n_rows + (uint64_t)TR * PAT_TILES - 1) / ((uint64_t)TR * PAT_TILES))), dim3(TR), lds, st, src, dmask, n_rows, S,
                       W_m, min_count, out, out_count);
    return hipGetLastError();
}

// Sorts keys[0..n) (destroys them) and returns the number of distinct values.
hipError_t count_distinct_u64(uint64_t* keys, uint64_t n, uint64_t* result, hipStream_t st) {
    *result = 0;
    if (n == 0) return hipSuccess;
    uint64_t* sorted = nullptr;
    void* temp = nullptr;
    unsigned long long* d_cnt = nullptr;
    size_t temp_bytes = 0;
    hipError_t e = hipMalloc((void**)&sorted, n * sizeof(uint64_t));
    if (e != hipSuccess) return e;
    auto cleanup = [&]() {
        if (sorted) (void)hipFree(sorted);
        if (temp) (void)hipFree(temp);
        if (d_cnt) (void)hipFree(d_cnt);
    };
    if ((e = hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys, sorted, n, 0, 64, st)) != hipSuccess ||
        (e = hipMalloc(&temp, temp_bytes ? temp_bytes : 8)) != hipSuccess ||
        (e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, sorted, n, 0, 64, st)) != hipSuccess ||
        (e = hipMalloc((void**)&d_cnt, 8)) != hipSuccess || (e = hipMemsetAsync(d_cnt, 0, 8, st)) != hipSuccess) {
        cleanup();
        return e;
    }
    hipLaunchKernelGGL(count_distinct_sorted_kernel, dim3(2048), dim3(256), 0, st, sorted, n, d_cnt);
    unsigned long long h = 0;
    if ((e = hipMemcpyAsync(&h, d_cnt, 8, hipMemcpyDeviceToHost, st)) == hipSuccess) e = hipStreamSynchronize(st);
    *result = h;
    cleanup();
    return e;
}

}  // namespace kgwas
