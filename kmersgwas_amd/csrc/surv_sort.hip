// surv_sort.hip — put the coarse filter's survivor keys in (column, row) order on the device.
//
// BestAssociationsHeap::add_association (src/best_associations_heap.cpp:43-59) is order dependent, so the
// host replays a column's candidates in file-row order. The coarse kernel appends keys
// (column << row_bits | chunk-local row) to one global list in arbitrary order; one radix sort of the list
// groups them by column and orders them by row, a tiny kernel finds each column's range, and the re-score kernel
// then emits its records already ordered, with coalesced writes: the host replay is a single forward scan.
#include <hipcub/hipcub.hpp>

#include "kernels.h"

namespace kgwas {

namespace {
// off[p] = first sorted key of column p, cnt[p] = how many (keys beyond n_keys are the 0xFFFFFFFF fill).
__global__ void surv_ranges_kernel(const uint32_t* keys, uint32_t n_slots, const uint32_t* key_count, uint32_t key_cap,
                                   uint32_t n_pheno, uint32_t row_bits, uint32_t* off, uint32_t* cnt) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pheno) return;
    uint32_t n = *key_count;
    if (n > key_cap) n = key_cap;  // the list overflowed: the host redoes the chunk (it reads key_count too)
    if (n > n_slots) n = n_slots;
    auto lower = [&](uint64_t v) {  // first index with key >= v
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)keys[mid] < v) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const uint32_t a = lower((uint64_t)p << row_bits), b = lower((uint64_t)(p + 1) << row_bits);
    off[p] = a;
    cnt[p] = b - a;
}
}  // namespace

hipError_t surv_sort_temp_bytes(uint32_t n_slots, size_t* bytes) {
    *bytes = 0;
    const uint32_t* kin = nullptr;
    uint32_t* kout = nullptr;
    return hipcub::DeviceRadixSort::SortKeys(nullptr, *bytes, kin, kout, (int)n_slots, 0, 32, 0);
}

hipError_t launch_surv_sort(const uint32_t* keys, uint32_t* keys_sorted, uint32_t n_slots, const uint32_t* key_count,
                            uint32_t key_cap, uint32_t n_pheno, uint32_t row_bits, uint32_t key_bits, uint32_t* off,
                            uint32_t* cnt, void* temp, size_t temp_bytes, hipStream_t st) {
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 32) key_bits = 32;
    // the whole slot array is sorted (its tail is the 0xFFFFFFFF fill): the number of live keys is only known on
    // the device, and a few million 32-bit keys sort in well under 0.1 ms
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, keys_sorted, (int)n_slots, 0, 32, st);
    if (e != hipSuccess) return e;
    (void)key_bits;
    hipLaunchKernelGGL(surv_ranges_kernel, dim3((n_pheno + 127u) / 128u), dim3(128), 0, st, keys_sorted, n_slots, key_count,
                       key_cap, n_pheno, row_bits, off, cnt);
    return hipGetLastError();
}

}  // namespace kgwas
