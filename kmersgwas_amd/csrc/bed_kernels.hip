// bed_kernels.hip — device side of kmers_table_to_bed.
//
// The reference loads a batch (MAC filter + per-bit squeeze, src/kmers_multiple_databases.cpp:103-147), hashes each
// kept row's presence/absence pattern when -u is given (:367-375) and writes one PLINK byte per 4 accessions
// (write_PA, :218-239). Here a chunk of table rows is squeezed on the device (squeeze_kernel), one kernel gives
// every row its masked popcount and pattern hash, another turns the squeezed bits into the .bed bytes (each
// presence bit doubled: b |= b << 1), and the host only filters, de-duplicates and writes in file order.
#include "kernels.h"

namespace kgwas {

namespace {

__device__ __forceinline__ uint64_t hash64_bed(uint64_t key) {  // Hash64, src/kmer_general.h:31-40
    key ^= key >> 33;
    key *= 0xff51afd7ed558ccdULL;
    key ^= key >> 33;
    key *= 0xc4ceb9fe1a85ec53ULL;
    key ^= key >> 33;
    return key;
}

// One lane per row over squeezed rows (2*W_m dwords each, zero padded): N1 and hash_presence_absence_pattern.
__global__ void __launch_bounds__(256) bed_rowinfo_kernel(const uint32_t* sq, uint64_t n_rows, uint32_t W_m, uint32_t* n1_out,
                                                          uint64_t* hash_out) {
    extern __shared__ uint32_t lds_rows[];
    const uint32_t ndw = 2u * W_m, ldw = ndw + 1u, TR = blockDim.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * TR;
    for (uint32_t e = threadIdx.x; e < TR * ndw; e += TR) {
        const uint32_t rr = e / ndw, dw = e - rr * ndw;
        const uint64_t gr = row0 + rr;
        lds_rows[rr * ldw + dw] = gr < n_rows ? sq[gr * ndw + dw] : 0u;
    }
    __syncthreads();
    const uint64_t r = row0 + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t* my = lds_rows + threadIdx.x * ldw;
    uint32_t n1 = 0;
    uint64_t seed = 0;
    for (uint32_t w = 0; w < W_m; w++) {
        const uint64_t word = (uint64_t)my[2 * w] | ((uint64_t)my[2 * w + 1] << 32);
        n1 += __popcll(word);
        seed ^= hash64_bed(word) + 0x9e3779b97f4a7c15ULL + (seed << 6) + (seed >> 2);
    }
    n1_out[r] = n1;
    hash_out[r] = seed;
}

// One lane per output byte: byte b of row r covers accessions 4b .. 4b+3 (bit pairs 00 / 11).
__global__ void __launch_bounds__(256) bed_bytes_kernel(const uint32_t* sq, uint64_t n_rows, uint32_t W_m, uint32_t bytes_per_row,
                                                        uint8_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * bytes_per_row) return;
    const uint64_t r = i / bytes_per_row;
    const uint32_t b = (uint32_t)(i - r * bytes_per_row);
    const uint32_t nib = (sq[r * 2u * W_m + (b >> 3)] >> ((b & 7u) * 4u)) & 0xFu;
    const uint32_t spread = (nib & 1u) | ((nib & 2u) << 1) | ((nib & 4u) << 2) | ((nib & 8u) << 3);
    out[i] = (uint8_t)(spread | (spread << 1));
}

}  // namespace

hipError_t launch_bed_rowinfo(const uint32_t* sq, uint64_t n_rows, uint32_t W_m, uint32_t* n1_out, uint64_t* hash_out,
                              hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    const uint32_t ldw = 2u * W_m + 1u;
    uint32_t TR = 256;
    while (TR > 64 && (size_t)TR * ldw * 4u > 150u * 1024u) TR >>= 1;
    const size_t lds = (size_t)TR * ldw * 4u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)bed_rowinfo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(bed_rowinfo_kernel, dim3((uint32_t)((n_rows + TR - 1) / TR)), dim3(TR), lds, st, sq, n_rows, W_m, n1_out,
                       hash_out);
    return hipGetLastError();
}

hipError_t launch_bed_bytes(const uint32_t* sq, uint64_t n_rows, uint32_t W_m, uint32_t bytes_per_row, uint8_t* out,
                            hipStream_t st) {
    const uint64_t n = n_rows * bytes_per_row;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bed_bytes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, sq, n_rows, W_m, bytes_per_row, out);
    return hipGetLastError();
}

}  // namespace kgwas
