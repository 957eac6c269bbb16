// score_common.h — device-side pieces shared by the two exact-order scoring kernels.
#pragma once
#include "kernels.h"

namespace kgwas {

// :119  (popcnt >= mac) && (popcnt <= accessions - mac), size_t arithmetic
// (src/kmers_multiple_databases.cpp:119; the same predicate guards calculate_kmer_score, :333-336)
__device__ __forceinline__ bool mac_pass(const ScoreArgs& a, uint32_t n1) {
    return (a.S >= a.min_count) && (n1 >= a.min_count) && (n1 <= a.S - a.min_count);
}

// The double-precision tail of calculate_kmer_score (src/kmers_multiple_databases.cpp:359-361)
// and the hand-off to the host-side BestAssociationsHeap replay.
//   dense mode : every score is written (0 for rows the MAC filter drops).
//   sparse mode: a record is shipped only if score > thr, thr being a stale value of the heap
//                minimum (add_association is a no-op for everything else once the heap is full,
//                src/best_associations_heap.cpp:49-58).
// sum_p / thr_p are the caller's register copies of sums[p] / thr[p].
__device__ __forceinline__ void finish_pair(const ScoreArgs& a, uint64_t r, uint32_t p, float yf, uint32_t n1,
                                            bool pass, float sum_p, double thr_p) {
    const double N = (double)a.S;
    const double N1 = (double)n1;
    const double yigi = (double)yf;
    const double rr = __dsub_rn(__dmul_rn(N, yigi), __dmul_rn(N1, (double)sum_p));  // no FMA (Makefile:4)
    const double q = __dmul_rn(rr, rr);
    const double d = __dsub_rn(__dmul_rn(N, N1), __dmul_rn(N1, N1));  // exact integers
    if (a.dense) {
        a.dense[(uint64_t)p * a.n_rows + r] = pass ? (q / d) : 0.0;
        return;
    }
    if (!pass) return;
    // Conservative prefilter without the division: fl(q/d) > t implies q > t*d >= lim.
    double lim = __dmul_rn(thr_p, d);
    lim = __dsub_rn(lim, __dmul_rn(fabs(lim), 0x1p-40));
    if (q >= lim) {
        const double s = q / d;  // correctly rounded IEEE division, as divsd on the host
        if (s > thr_p) {
            if (a.hist) {
                // Count the score in its bin (positive doubles order like their bit patterns; s > thr_p >= 0
                // here, +inf lands in the clamped top bin, NaN never gets here).
                const uint32_t b = (uint32_t)((unsigned long long)__double_as_longlong(s) >> HIST_SHIFT);
                const uint32_t base = a.hist_base[p];
                uint32_t idx = b > base ? b - base : 0u;
                if (idx >= a.hist_bins) idx = a.hist_bins - 1u;
                atomicAdd(&a.hist[(uint64_t)p * a.hist_bins + idx], 1u);
            }
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

}  // namespace kgwas
