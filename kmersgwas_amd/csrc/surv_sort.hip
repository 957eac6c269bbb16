// surv_sort.hip — put the coarse filter's survivor lists in row order on the device.
//
// BestAssociationsHeap::add_association (src/best_associations_heap.cpp:43-59) is order dependent, so the
// host replays a column's candidates in file-row order. The coarse kernel appends survivors with atomics
// (arbitrary order); sorting the chunk-local row indices here (one segment per phenotype column, keys are
// distinct) lets the re-score kernel emit its records already ordered, with coalesced writes, and the host
// replay becomes a single forward scan.
#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace kgwas {

namespace {
__global__ void seg_end_kernel(const uint32_t* cnt, uint32_t cap, uint32_t n_pheno, uint32_t* seg_end) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pheno) return;
    const uint32_t n = cnt[p];
    seg_end[p] = p * cap + (n > cap ? 0u : n);  // an overflowing list is redone by the host anyway: skip it
}
__global__ void seg_begin_kernel(uint32_t cap, uint32_t n_pheno, uint32_t* seg_beg) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pheno) seg_beg[p] = p * cap;
}
}  // namespace

hipError_t surv_sort_temp_bytes(uint32_t n_pheno, uint32_t cap, size_t* bytes) {
    *bytes = 0;
    const uint32_t* kin = nullptr;
    uint32_t* kout = nullptr;
    const uint32_t* off = nullptr;
    return hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, *bytes, kin, kout, (int)((uint64_t)n_pheno * cap),
                                                      (int)n_pheno, off, off, 0, 32, 0);
}

hipError_t launch_seg_begin(uint32_t cap, uint32_t n_pheno, uint32_t* seg_beg, hipStream_t st) {
    hipLaunchKernelGGL(seg_begin_kernel, dim3((n_pheno + 255u) / 256u), dim3(256), 0, st, cap, n_pheno, seg_beg);
    return hipGetLastError();
}

hipError_t launch_surv_sort(const uint32_t* surv, uint32_t* surv_sorted, const uint32_t* surv_cnt, const uint32_t* seg_beg,
                            uint32_t* seg_end, uint32_t n_pheno, uint32_t cap, uint32_t key_bits, void* temp,
                            size_t temp_bytes, hipStream_t st) {
    hipLaunchKernelGGL(seg_end_kernel, dim3((n_pheno + 255u) / 256u), dim3(256), 0, st, surv_cnt, cap, n_pheno, seg_end);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 32) key_bits = 32;
    return hipcub::DeviceSegmentedRadixSort::SortKeys(temp, temp_bytes, surv, surv_sorted, (int)((uint64_t)n_pheno * cap),
                                                      (int)n_pheno, seg_beg, (const uint32_t*)seg_end, 0, (int)key_bits, st);
}

}  // namespace kgwas
