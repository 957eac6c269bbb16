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

// ---- device-side threshold tracking (aux_kernels.hip describes it) -----------------------------------------------------
// One 256-thread block raises thr[p] (thr_update_kernel's body: one block per column there; the narrow scans run it in the
// first blocks of the NEXT chunk's prep kernel instead of a launch of its own). Ends with a block barrier.
// Thread t owns the segment of bins [t * per, (t + 1) * per); suffix sums over the 256 segments (Hillis-Steele from the
// top) find the segment that takes the count past N, and wave 0 then looks at that segment's bins side by side (one thread
// walking them was a chain of dependent loads: most of the 12 us this used to take).
__device__ __forceinline__ void thr_update_block(const uint32_t* hist, const uint32_t* hist_base, uint32_t bins, const uint64_t* topn,
                                                 const double* thr_host, double* thr, uint32_t p) {
    __shared__ unsigned long long suf[256];
    __shared__ int seg_s;
    __shared__ unsigned long long run_s;
    const uint32_t t = threadIdx.x;
    const uint32_t per = bins / 256u;  // 64 with HIST_BINS = 16384 (any multiple of 4 up to 64 works below)
    const uint32_t* h = hist + (uint64_t)p * bins;
    unsigned long long s = 0;
    {
        const uint4* h4 = reinterpret_cast<const uint4*>(h + (size_t)t * per);
        for (uint32_t i = 0; i < per / 4u; i++) {
            const uint4 v = h4[i];
            s += (unsigned long long)v.x + v.y + v.z + v.w;
        }
    }
    suf[t] = s;
    if (t == 0) seg_s = -1;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const unsigned long long x = t + d < 256u ? suf[t + d] : 0ull;
        __syncthreads();
        suf[t] += x;
        __syncthreads();
    }
    const unsigned long long N = topn[p];
    {
        const unsigned long long above = suf[t] - s;  // segments t + 1 .. 255
        if (above < N && above + s >= N) {  // exactly one t (or none: fewer than N scores counted)
            seg_s = (int)t;
            run_s = above;
        }
    }
    __syncthreads();
    if (t < 64u) {  // wave 0
        const int seg = seg_s;
        double cur = thr[p];
        const double th = thr_host[p];
        if (seg >= 0) {
            // lane l looks at bin b = seg * per + per - 1 - l (from the top down): incl = scores counted in the bins above the
            // segment and in the segment's bins b and higher; the first lane with incl >= N holds the boundary (the last
            // bin of the segment if none: cannot happen, above + s >= N)
            const uint32_t l = t;
            const uint32_t b = (uint32_t)seg * per + per - 1u - (l < per ? l : per - 1u);
            unsigned long long incl = l < per ? h[b] : 0ull;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned long long x = __shfl_up(incl, d);
                if ((int)l >= d) incl += x;
            }
            incl += run_s;
            const unsigned long long m = __ballot(l < per && incl >= N);
            const uint32_t first = m ? (uint32_t)__ffsll((long long)m) - 1u : per - 1u;
            const uint32_t bb = (uint32_t)seg * per + per - 1u - first;
            const double v = __longlong_as_double((long long)((unsigned long long)(hist_base[p] + bb) << HIST_SHIFT));
            if (v > cur) cur = v;
        }
        if (th != th || cur != cur)
            cur = __longlong_as_double(0x7FF8000000000000LL);
        else if (th > cur)
            cur = th;
        if (t == 0) thr[p] = cur;
    }
    __syncthreads();
}

}  // namespace kgwas
