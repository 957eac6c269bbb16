// score_common.h — device-side pieces shared by the two exact-order scoring kernels.
#pragma once
#include "kernels.h"

namespace kgwas {

// :119  (popcnt >= mac) && (popcnt <= accessions - mac), size_t arithmetic
// (src/kmers_multiple_databases.cpp:119; the same predicate guards calculate_kmer_score, :333-336)
__device__ __forceinline__ bool mac_pass(const ScoreArgs& a, uint32_t n1) {
    return (a.S >= a.min_count) && (n1 >= a.min_count) && (n1 <= a.S - a.min_count);
}

// Sparse mode: the double-precision tail of calculate_kmer_score (src/kmers_multiple_databases.cpp:359-361)
// for a MAC-passing pair, evaluated only as far as needed: returns true and the exact score in s iff
// score > thr_p (thr_p being a stale value of the heap minimum: add_association is a no-op for everything
// else once the heap is full, src/best_associations_heap.cpp:49-58). Counts the score in the column's
// threshold histogram when one is attached.
__device__ __forceinline__ bool candidate_score(const ScoreArgs& a, uint32_t p, double q, double d, double thr_p,
                                                double& s) {
    // Conservative prefilter without the division: fl(q/d) > t implies q > t*d >= lim.
    double lim = __dmul_rn(thr_p, d);
    lim = __dsub_rn(lim, __dmul_rn(fabs(lim), 0x1p-40));
    if (!(q >= lim)) return false;
    s = q / d;  // correctly rounded IEEE division, as divsd on the host
    if (!(s > thr_p)) return false;
    if (a.hist) {
        // Count the score in its bin (positive doubles order like their bit patterns; s > thr_p >= 0
        // here, +inf lands in the clamped top bin, NaN never gets here).
        const uint32_t b = (uint32_t)((unsigned long long)__double_as_longlong(s) >> HIST_SHIFT);
        const uint32_t base = a.hist_base[p];
        uint32_t idx = b > base ? b - base : 0u;
        if (idx >= a.hist_bins) idx = a.hist_bins - 1u;
        atomicAdd(&a.hist[(uint64_t)p * a.hist_bins + idx], 1u);
    }
    return true;
}

// r = N*yigi - N1*sum and the two factors of the score r*r / (N*N1 - N1*N1), exactly as the reference
// rounds them (no FMA, Makefile:4).
__device__ __forceinline__ void score_terms(const ScoreArgs& a, float yf, uint32_t n1, float sum_p, double& q, double& d) {
    const double N = (double)a.S;
    const double N1 = (double)n1;
    const double yigi = (double)yf;
    const double rr = __dsub_rn(__dmul_rn(N, yigi), __dmul_rn(N1, (double)sum_p));
    q = __dmul_rn(rr, rr);
    d = __dsub_rn(__dmul_rn(N, N1), __dmul_rn(N1, N1));  // exact integers
}

// Hand-off of one (row, column) pair to the host-side BestAssociationsHeap replay.
//   dense mode : every score is written (0 for rows the MAC filter drops).
//   sparse mode: a record is shipped only if score > thr.
// sum_p / thr_p are the caller's register copies of sums[p] / thr[p].
__device__ __forceinline__ void finish_pair(const ScoreArgs& a, uint64_t r, uint32_t p, float yf, uint32_t n1,
                                            bool pass, float sum_p, double thr_p) {
    double q, d;
    score_terms(a, yf, n1, sum_p, q, d);
    if (a.dense) {
        a.dense[(uint64_t)p * a.n_rows + r] = pass ? (q / d) : 0.0;
        return;
    }
    if (!pass) return;
    double s;
    if (candidate_score(a, p, q, d, thr_p, s)) {
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

}  // namespace kgwas
