// patterns.hip — `--pattern_counter`: the number of distinct presence/absence patterns among the
// MAC-passing rows (src/associate_kmers.cpp:113-115,130-132,197-201).
//
// The reference inserts hash_presence_absence_pattern(row) of every kept row into a dense_hash_set
// and reports the set's size (src/kmers_multiple_databases.cpp:367-380): the hash is a boost-style
// combine of Hash64 (src/kmer_general.h:31-40) over the squeezed, zero-padded W_m words. Here one
// kernel hashes the rows of a chunk (rows staged through LDS, one lane per row) and appends the hashes
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

__global__ void __launch_bounds__(256) pattern_hash_kernel(RowSrc src, const uint32_t* dmask, uint64_t n_rows, uint32_t S,
                                                           uint32_t W_m, uint32_t min_count, uint64_t* out,
                                                           unsigned long long* out_count) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t ndw = 2u * W_m;
    const uint32_t ldw = ndw + 1u;
    const uint32_t TR = blockDim.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * TR;
    for (uint32_t e = threadIdx.x; e < TR * ndw; e += TR) {
        const uint32_t rr = e / ndw, dw = e - rr * ndw;
        const uint64_t gr = row0 + rr;
        uint32_t v = 0;
        if (gr < n_rows && dw < src.avail_dw) v = src.base[gr * src.stride_dw + src.off_dw + dw] & dmask[dw];
        lds_u32[rr * ldw + dw] = v;
    }
    __syncthreads();
    const uint64_t r = row0 + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t* my = lds_u32 + threadIdx.x * ldw;
    uint32_t n1 = 0;
    uint64_t seed = 0;
    for (uint32_t w = 0; w < W_m; w++) {
        const uint64_t word = (uint64_t)my[2 * w] | ((uint64_t)my[2 * w + 1] << 32);
        n1 += __popcll(word);
        seed ^= hash64(word) + 0x9e3779b97f4a7c15ULL + (seed << 6) + (seed >> 2);
    }
    if (S >= min_count && n1 >= min_count && n1 <= S - min_count) out[atomicAdd(out_count, 1ull)] = seed;
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
    const uint32_t ldw = 2u * W_m + 1u;
    uint32_t TR = 256;
    while (TR > 64 && (size_t)TR * ldw * 4u > 150u * 1024u) TR >>= 1;
    const size_t lds = (size_t)TR * ldw * 4u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)pattern_hash_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(pattern_hash_kernel, dim3((uint32_t)((n_rows + TR - 1) / TR)), dim3(TR), lds, st, src, dmask, n_rows, S,
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
