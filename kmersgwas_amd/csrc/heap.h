// heap.h — BestHeap: the host-side mirror of the reference's BestAssociationsHeap
// (src/best_associations_heap.h:32-54, src/best_associations_heap.cpp:26-127).
//
// The reference keeps (k-mer, score, row) tuples in a std::priority_queue ordered by
// cmp(a, b) = a.score > b.score (src/kmer_general.h:113-128). Which of several equal-score
// entries survives at the boundary, and the order equal scores pop in (= the rank written
// into the .bim names), is decided by libstdc++'s push_heap / pop_heap and the exact history
// of effective pushes. std::priority_queue is nothing but a vector driven by exactly these calls
//     push: c.push_back(x); std::push_heap(c.begin(), c.end(), comp);
//     pop : std::pop_heap(c.begin(), c.end(), comp); c.pop_back();
//     top : c.front()
// and the algorithms only ever look at the elements through `comp`. We issue the same calls on a
// vector of 16-byte (score, slot) entries — k-mer and row live in a side array of 16-byte pairs indexed by slot
// (one cache line written per push, not two) — so
// every comparison, and therefore every move, is the one the reference's 24-byte tuples would
// make, at 2/3 of the memory traffic of the replay (the hot loop of the host side).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <memory_resource>
#include <vector>

namespace kgwas {

// alignas: neighbouring heaps are replayed by different worker threads; their counters and minima are written
// on every call, so each object gets its own cache lines (no false sharing).
class alignas(128) BestHeap {
    struct Ent {
        double score;
        uint32_t slot;
    };
    struct Pay {  // what an entry carries besides its score
        uint64_t kmer, row;
    };
    struct Greater {
        inline bool operator()(const Ent& l, const Ent& r) const { return l.score > r.score; }
    };
    struct GreaterInt {
        inline bool operator()(const Ent& l, const Ent& r) const { return gt<true>(l, r); }
    };
    // l.score > r.score. INT: on the doubles' bit patterns as int64 - the same answer for non-negative, non-NaN
    // scores (IEEE order is integer order there; ints_ok_ tracks that every score ever inserted was one), and the
    // compare a hole walk's next address waits for costs an integer load + cmp (4 + 1 cycles) instead of a load into
    // the FP domain + ucomisd (7 + 3): 55 -> 40 ns per push for a single heap, 26 -> 23.6 in 7-way lockstep
    // (tools/heap_soa_bench.cpp, identical layouts).
    template <bool INT>
    static inline bool gt(const Ent& l, const Ent& r) {
        if (INT) {
            int64_t a, b;
            memcpy(&a, &l.score, 8);
            memcpy(&b, &r.score, 8);
            return a > b;
        }
        return l.score > r.score;
    }
    static inline bool int_comparable(double score) {  // +0 .. +inf
        uint64_t b;
        memcpy(&b, &score, 8);
        return b <= 0x7FF0000000000000ull;
    }

   public:
    // mr: where the entry and payload arrays live (a scan session packs all its heaps into one huge-page arena - a
    // heap walk touches 13 levels of a 160 KB array and a random 16-byte payload slot, and with 4 KB pages the dTLB
    // misses cost 5-7 % of a push; tools/heap_soa_bench.cpp). With a resource both arrays are reserved in full up front.
    explicit BestHeap(size_t max_results, std::pmr::memory_resource* mr = nullptr)
        : n_res_(max_results), v_(mr ? mr : std::pmr::get_default_resource()), pay_(mr ? mr : std::pmr::get_default_resource()),
          inserted_(0), pushes_(0), lowest_(0) {
        if (mr) {
            v_.reserve(max_results);
            pay_.reserve(max_results);
        }
    }

    // add_association (src/best_associations_heap.cpp:43-59). Returns true if the heap changed.
    inline bool add(uint64_t kmer, double score, size_t row) {
        inserted_++;
        if (v_.size() < n_res_) {
            ints_ok_ = ints_ok_ && int_comparable(score);
            const uint32_t slot = (uint32_t)v_.size();
            pay_.push_back(Pay{kmer, (uint64_t)row});
            v_.push_back(Ent{score, slot});
            if (ints_ok_)
                push_up<true>(v_.data(), (ptrdiff_t)v_.size() - 1, Ent{score, slot});  // (same answers, hence the same moves)
            else
                push_up<false>(v_.data(), (ptrdiff_t)v_.size() - 1, Ent{score, slot});
            pushes_++;
            lowest_ = v_.front().score;
            return true;
        }
        if (score > lowest_) {
            const uint32_t slot = v_.front().slot;  // the evicted minimum's slot is reused
            note_eviction(slot, v_.front().score);
            pay_[slot] = Pay{kmer, (uint64_t)row};
            ints_ok_ = ints_ok_ && int_comparable(score);
            if (ints_ok_)
                replace_top<true>(v_.data(), (ptrdiff_t)v_.size(), Ent{score, slot});
            else
                replace_top<false>(v_.data(), (ptrdiff_t)v_.size(), Ent{score, slot});
            pushes_++;
            lowest_ = v_.front().score;
            return true;
        }
        return false;
    }

    // std::push_heap's moves for the element x that was just appended at a[hole] (libstdc++'s __push_heap: x climbs
    // while its parent compares greater, parents move down, a tie stops the climb). While a heap fills this runs once per
    // row and column (1 M times per scan at 101 columns, on the scan's serial start), and how far x climbs is a coin flip
    // per level for the branch predictor: the first three levels are taken without branches (three independent loads,
    // selects, three stores that rewrite what is already there when x stays below), the rest - one push in eight - by
    // the plain loop. 31 -> 17 ns per push (tests/test_host.py drives it against a literal std::priority_queue).
    template <bool INT>
    static inline void push_up(Ent* a, ptrdiff_t hole, Ent x) {
        if (hole >= 16) {
            const ptrdiff_t p1 = (hole - 1) >> 1, p2 = (p1 - 1) >> 1, p3 = (p2 - 1) >> 1;
            const Ent e1 = a[p1], e2 = a[p2], e3 = a[p3];
            const bool u1 = gt<INT>(e1, x), u2 = u1 & gt<INT>(e2, x), u3 = u2 & gt<INT>(e3, x);
            a[hole] = sel(u1, e1, x);
            a[p1] = sel(u2, e2, sel(u1, x, e1));
            a[p2] = sel(u3, e3, sel(u2, x, e2));
            if (__builtin_expect(!u3, 1)) return;
            hole = p3;
        }
        ptrdiff_t parent = (hole - 1) / 2;
        while (hole > 0 && gt<INT>(a[parent], x)) {
            a[hole] = a[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[hole] = x;
    }
    static inline Ent sel(bool c, const Ent& t, const Ent& f) {  // c ? t : f on the two 64-bit halves (selects, not a branch)
        uint64_t t0, t1, f0, f1;
        memcpy(&t0, &t.score, 8);
        memcpy(&f0, &f.score, 8);
        t1 = t.slot;
        f1 = f.slot;
        const uint64_t m = (uint64_t)0 - (uint64_t)c;
        const uint64_t r0 = (t0 & m) | (f0 & ~m);
        Ent r;
        memcpy(&r.score, &r0, 8);
        r.slot = (uint32_t)((t1 & m) | (f1 & ~m));
        return r;
    }

    // pop() followed by push(x) on a full heap a[0..n), written out: the element moves are exactly those of
    // libstdc++'s std::pop_heap (value = a[n-1]; __adjust_heap: the hole at the root walks down to a leaf
    // taking the child that does NOT compare greater, the right one on a tie; __push_heap: value climbs from
    // that leaf while its parent compares greater) and std::push_heap (x climbs from index n-1), in that
    // order, with the comparator Greater. Only the child choice is made branch-free (it is a coin flip for
    // the branch predictor, 13 levels deep at N = 10001). tests/test_host.py drives this against a literal
    // std::priority_queue on tie-heavy streams.
    // Heaps that do not fit the cache (the reference's default -n 1 000 000: 16 MB of entries): the hole walk is a chain of
    // dependent misses, 20 levels deep. Which way it turns is not known in advance, but WHERE it can be three levels down is:
    // the 16 descendants of node c at depth + 4 are contiguous (16 c + 15 .. 16 c + 30, five cache lines); asking for them while
    // the compares of the levels in between are made turns every miss but the first into a hit in flight
    // (tools/heap_one_bench.cpp, N = 10^6). No effect on what is moved where.
    static constexpr ptrdiff_t BIG_HEAP = 1 << 16;
    static inline void prefetch_below(const Ent* a, ptrdiff_t c, ptrdiff_t n) {
        const ptrdiff_t d = 16 * c + 15;
        if (d + 15 < n) {
            __builtin_prefetch(a + d);
            __builtin_prefetch(a + d + 4);
            __builtin_prefetch(a + d + 8);
            __builtin_prefetch(a + d + 12);
            __builtin_prefetch(a + d + 15);
        }
    }
    template <bool INT>
    static inline void replace_top(Ent* a, ptrdiff_t n, Ent x) {
        const Ent value = a[n - 1];
        const ptrdiff_t len = n - 1;
        ptrdiff_t hole = 0, child = 0;
        if (INT) {
            // Totally ordered scores: the keys on the hole's path never decrease downwards, so "walk to a leaf, then climb
            // while the parent is greater than `value`" ends at the first path node whose key exceeds `value` - the hole
            // stops there instead (same final array; one heap at a time this saves the last level or two and the climb:
            // 40 -> 34 ns per push, tools/heap_soa_bench.cpp; in lockstep the extra compare per level costs more).
            bool open = true;
            const bool big = n > BIG_HEAP;
            while (child < (len - 1) / 2) {
                if (big) prefetch_below(a, child, n);
                ptrdiff_t cc = 2 * (child + 1);
                cc -= gt<true>(a[cc], a[cc - 1]) ? 1 : 0;
                if (gt<true>(a[cc], value)) {
                    open = false;
                    break;
                }
                a[hole] = a[cc];
                hole = child = cc;
            }
            if (open && (len & 1) == 0 && child == (len - 2) / 2) {  // the last inner node has a left child only
                const ptrdiff_t c2 = 2 * (child + 1);
                if (!gt<true>(a[c2 - 1], value)) {
                    a[hole] = a[c2 - 1];
                    hole = c2 - 1;
                }
            }
            a[hole] = value;
            hole = n - 1;
            ptrdiff_t parent = (hole - 1) / 2;
            while (hole > 0 && gt<true>(a[parent], x)) {
                a[hole] = a[parent];
                hole = parent;
                parent = (hole - 1) / 2;
            }
            a[hole] = x;
            return;
        }
        const bool big = n > BIG_HEAP;
        while (child < (len - 1) / 2) {
            if (big) prefetch_below(a, child, n);
            child = 2 * (child + 1);
            child -= gt<INT>(a[child], a[child - 1]) ? 1 : 0;
            a[hole] = a[child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            a[hole] = a[child - 1];
            hole = child - 1;
        }
        ptrdiff_t parent = (hole - 1) / 2;
        while (hole > 0 && gt<INT>(a[parent], value)) {
            a[hole] = a[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[hole] = value;
        hole = n - 1;
        parent = (hole - 1) / 2;
        while (hole > 0 && gt<INT>(a[parent], x)) {
            a[hole] = a[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[hole] = x;
    }
    inline bool full() const { return v_.size() >= n_res_; }
    inline size_t size() const { return v_.size(); }
    inline size_t capacity() const { return n_res_; }
    inline double lowest() const { return lowest_; }
    inline uint64_t inserted() const { return inserted_; }
    inline uint64_t pushes() const { return pushes_; }

    // The same operation on K full heaps of EQUAL size, advanced in lockstep inside one loop. Each heap's element
    // moves are exactly those of replace_top; interleaving K independent walks only gives the core K dependency
    // chains to overlap (one walk is ~13 dependent load-compare-select steps): 29 ns per heap at K = 4 against
    // 53 ns one at a time on an EPYC 9575F (tools/heap_bench.cpp, which also checks that the layouts are identical).
    // Preconditions: every heap full, same size(), score[k] > hp[k]->lowest().
    template <int K, bool INT>
    static inline void replace_top_multi(BestHeap* const* hp, const uint64_t* kmer, const double* score,
                                         const uint64_t* row) {
        Ent* a[K];
        Ent x[K], v[K];
        ptrdiff_t h[K], c[K];
        const ptrdiff_t n = (ptrdiff_t)hp[0]->v_.size();
        const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
#pragma unroll
        for (int k = 0; k < K; k++) {
            BestHeap& H = *hp[k];
            a[k] = H.v_.data();
            const uint32_t slot = a[k][0].slot;
            H.note_eviction(slot, a[k][0].score);
            H.pay_[slot] = Pay{kmer[k], row[k]};
            x[k] = Ent{score[k], slot};
            v[k] = a[k][n - 1];
            h[k] = 0;
            c[k] = 0;
            H.inserted_++;
            H.pushes_++;
        }
        if (K == 1 && INT) {  // one heap: the early-stopping form
            replace_top<true>(a[0], n, x[0]);
            hp[0]->lowest_ = a[0][0].score;
            return;
        }
        const bool big = n > BIG_HEAP;
        for (;;) {  // hole walks: all reach the leaf level within one step of each other
            bool any = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (c[k] < lim) {
                    if (big) prefetch_below(a[k], c[k], n);
                    ptrdiff_t cc = 2 * (c[k] + 1);
                    cc -= gt<INT>(a[k][cc], a[k][cc - 1]) ? 1 : 0;
                    a[k][h[k]] = a[k][cc];
                    h[k] = cc;
                    c[k] = cc;
                    any = true;
                }
            }
            if (!any) break;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
                c[k] = 2 * (c[k] + 1);
                a[k][h[k]] = a[k][c[k] - 1];
                h[k] = c[k] - 1;
            }
            ptrdiff_t hh = h[k], p = (hh - 1) / 2;
            while (hh > 0 && gt<INT>(a[k][p], v[k])) {
                a[k][hh] = a[k][p];
                hh = p;
                p = (hh - 1) / 2;
            }
            a[k][hh] = v[k];
            hh = n - 1;
            p = (hh - 1) / 2;
            while (hh > 0 && gt<INT>(a[k][p], x[k])) {
                a[k][hh] = a[k][p];
                hh = p;
                p = (hh - 1) / 2;
            }
            a[k][hh] = x[k];
            hp[k]->lowest_ = a[k][0].score;
        }
    }
    static constexpr int MAX_LOCKSTEP = 8;
    static inline void replace_top_n(int K, BestHeap* const* hp, const uint64_t* kmer, const double* score,
                                     const uint64_t* row) {
        bool ints = true;  // every heap of the group, and every new score
        for (int k = 0; k < K; k++) {
            hp[k]->ints_ok_ = hp[k]->ints_ok_ && int_comparable(score[k]);
            ints = ints && hp[k]->ints_ok_;
        }
        if (ints) {
            switch (K) {
                case 1: replace_top_multi<1, true>(hp, kmer, score, row); break;
                case 2: replace_top_multi<2, true>(hp, kmer, score, row); break;
                case 3: replace_top_multi<3, true>(hp, kmer, score, row); break;
                case 4: replace_top_multi<4, true>(hp, kmer, score, row); break;
                case 5: replace_top_multi<5, true>(hp, kmer, score, row); break;
                case 6: replace_top_multi<6, true>(hp, kmer, score, row); break;
                case 7: replace_top_multi<7, true>(hp, kmer, score, row); break;
                default: replace_top_multi<8, true>(hp, kmer, score, row); break;
            }
            return;
        }
        switch (K) {
            case 1: replace_top_multi<1, false>(hp, kmer, score, row); break;
            case 2: replace_top_multi<2, false>(hp, kmer, score, row); break;
            case 3: replace_top_multi<3, false>(hp, kmer, score, row); break;
            case 4: replace_top_multi<4, false>(hp, kmer, score, row); break;
            case 5: replace_top_multi<5, false>(hp, kmer, score, row); break;
            case 6: replace_top_multi<6, false>(hp, kmer, score, row); break;
            case 7: replace_top_multi<7, false>(hp, kmer, score, row); break;
            default: replace_top_multi<8, false>(hp, kmer, score, row); break;
        }
    }
    // The heap's state in heap-array order (entry i = array position i): shipping these three arrays and importing
    // them elsewhere reproduces the layout exactly, so every later push/pop behaves as it would have here
    // (multi-GPU merge: column j continues on the rank that owns it).
    void export_state(uint64_t* kmer, double* score, uint64_t* row) const {  // size() entries each
        size_t i = 0;
        for (const Ent& e : v_) {
            kmer[i] = pay_[e.slot].kmer;
            score[i] = e.score;
            row[i] = pay_[e.slot].row;
            i++;
        }
    }
    void import_state(size_t n, const uint64_t* kmer, const double* score, const uint64_t* row) {
        if (n > n_res_) n = n_res_;
        v_.clear();
        pay_.clear();
        evicted_ = 0;
        ints_ok_ = true;
        for (size_t i = 0; i < n; i++) {
            pay_.push_back(Pay{kmer[i], row[i]});
            v_.push_back(Ent{score[i], (uint32_t)i});
            ints_ok_ = ints_ok_ && int_comparable(score[i]);
        }
        lowest_ = n ? v_.front().score : 0;
    }

    // add_association for a record that is known not to beat a full heap's minimum: only the call counter moves.
    inline void note_rejected(uint64_t n = 1) { inserted_ += n; }

    // output_to_file_with_scores order (:82-92): ascending pops from a copy.
    // The pop sequence WITHOUT popping, where the scores alone decide it: if every score is +0 .. +inf (ints_ok_) and no
    // two entries share one, N pops yield the entries in ascending score order whatever the array's layout - a heap's pop
    // order can only depend on the layout among EQUAL scores. The entries' bit patterns are radix-sorted (11-bit digits,
    // digits in which all keys agree skipped: the top-N of a scan share sign, exponent and often the leading mantissa
    // bits) and neighbours compared; on the first equal pair the function gives up (false, outputs untouched) and the
    // caller pops for real. 10 001 entries: ~0.1 ms against 0.5 ms of pops alone, 0.27 in 8-way lockstep.
    // last_tie (optional): where a tie stops the function, the index - in ascending order - of the LAST entry that equals its
    // predecessor: everything behind it is distinct, and pop_all pops only up to it for real.
    bool pop_all_sorted(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row, size_t* last_tie = nullptr) const {
        if (!ints_ok_) return false;
        const size_t n = v_.size();
        struct Key {
            uint64_t bits;
            uint32_t slot;
        };
        // (scratch kept per thread: two fresh 160 KB vectors per call are two mmaps' worth of page faults, a third of the call)
        static thread_local std::vector<Key> a, b;
        if (a.size() < n) a.resize(n), b.resize(n);
        uint64_t all_or = 0, all_and = ~0ull;
        for (size_t i = 0; i < n; i++) {
            uint64_t x;
            memcpy(&x, &v_[i].score, 8);
            a[i] = Key{x, v_[i].slot};
            all_or |= x;
            all_and &= x;
        }
        const uint64_t varying = all_or ^ all_and;  // bits in which at least two keys differ
        Key* src = a.data();
        Key* dst = b.data();
        for (int shift = 0; shift < 64; shift += 11) {
            if (((varying >> shift) & 0x7FFull) == 0) continue;
            uint32_t cnt[2048] = {0};
            for (size_t i = 0; i < n; i++) cnt[(src[i].bits >> shift) & 0x7FFu]++;
            uint32_t run = 0;
            for (int d = 0; d < 2048; d++) {
                const uint32_t c = cnt[d];
                cnt[d] = run;
                run += c;
            }
            for (size_t i = 0; i < n; i++) dst[cnt[(src[i].bits >> shift) & 0x7FFu]++] = src[i];
            std::swap(src, dst);
        }
        size_t tie_at = 0;  // the last i with src[i] == src[i - 1]
        for (size_t i = n; i-- > 1;)
            if (src[i].bits == src[i - 1].bits) {
                tie_at = i;
                break;
            }
        const size_t first = tie_at ? tie_at + 1 : 0;  // entries [first, n) are distinct and above every tied score
        if (tie_at && (!last_tie || first > n - n / 8)) return false;  // a tie: its pop order is the layout's business
        kmer.resize(n);
        score.resize(n);
        row.resize(n);
        for (size_t i = first; i < n; i++) {
            memcpy(&score[i], &src[i].bits, 8);
            kmer[i] = pay_[src[i].slot].kmer;
            row[i] = pay_[src[i].slot].row;
        }
        if (tie_at) {
            *last_tie = tie_at;
            return false;  // (outputs [first, n) are filled; the caller pops the first `first` entries for real)
        }
        return true;
    }

    // Pops are ascending in score whatever the layout; only the order of EQUAL scores is the layout's. So a heap with ties is
    // popped for real just until its last tied score has come out - the entries above it are distinct, and their pop order is
    // their sorted order (a column whose tie turned up late in a scan is replayed and popped at the scan's very end: the pair that
    // ties sits anywhere among its 10 001 entries, half the pops go on average).
    void pop_all(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row) const {
        size_t last_tie = 0;
        if (pop_all_sorted(kmer, score, row, &last_tie)) return;
        if (!last_tie) {
            pop_all_classic(kmer, score, row);
            return;
        }
        // outputs [last_tie + 1, n) are in place; the first last_tie + 1 pops on a copy of the array
        std::vector<Ent> tmp(v_.begin(), v_.end());
        const size_t n = tmp.size();
        Ent* a = tmp.data();
        for (size_t i = 0; i <= last_tie; i++) {
            kmer[i] = pay_[a[0].slot].kmer;
            score[i] = a[0].score;
            row[i] = pay_[a[0].slot].row;
            pop_top<true>(a, (ptrdiff_t)(n - i));
        }
    }
    // get_rows_sorted_indices (src/best_associations_heap.cpp:135-147): the entries' rows, ascending
    std::vector<uint64_t> rows_sorted() const {
        std::vector<uint64_t> r(pay_.size());
        for (size_t i = 0; i < pay_.size(); i++) r[i] = pay_[i].row;
        std::sort(r.begin(), r.end());
        return r;
    }
    // N pops on a copy of the array, with std::pop_heap's moves
    void pop_all_classic(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row) const {
        std::vector<Ent> tmp(v_.begin(), v_.end());
        const size_t n = tmp.size();
        kmer.resize(n);
        score.resize(n);
        row.resize(n);
        Ent* a = tmp.data();
        for (size_t i = 0; i < n; i++) {
            kmer[i] = pay_[a[0].slot].kmer;
            score[i] = a[0].score;
            row[i] = pay_[a[0].slot].row;
            if (ints_ok_) pop_top<true>(a, (ptrdiff_t)(n - i)); else pop_top<false>(a, (ptrdiff_t)(n - i));
        }
    }

    // std::pop_heap(a, a + n, Greater()) minus the store of the old top into a[n-1] (the caller drops it):
    // same hole walk and climb as in replace_top.
    template <bool INT>
    static inline void pop_top(Ent* a, ptrdiff_t n) {
        if (n <= 1) return;
        const Ent value = a[n - 1];
        const ptrdiff_t len = n - 1;
        ptrdiff_t hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            child -= gt<INT>(a[child], a[child - 1]) ? 1 : 0;
            a[hole] = a[child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            a[hole] = a[child - 1];
            hole = child - 1;
        }
        ptrdiff_t parent = (hole - 1) / 2;
        while (hole > 0 && gt<INT>(a[parent], value)) {
            a[hole] = a[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[hole] = value;
    }

    // pop_all for K heaps of equal size() at once: the K pop sequences advance in lockstep (same idea as
    // replace_top_multi; each heap's moves are those of pop_top).
    // K heaps of equal size() (1 <= K <= 8): pop_all of each, in lockstep; integer compares if all of them allow it
    static void pop_all_n(int K, const BestHeap* const* hp, std::vector<uint64_t>* const* kmer, std::vector<double>* const* score,
                          std::vector<uint64_t>* const* row) {
        {  // heaps whose scores decide the order are sorted, not popped; the rest (ties, NaN, negative scores) go on in lockstep
            const BestHeap* rest_h[8];
            std::vector<uint64_t>* rest_k[8];
            std::vector<double>* rest_s[8];
            std::vector<uint64_t>* rest_r[8];
            int R = 0;
            for (int k = 0; k < K; k++)
                if (!hp[k]->pop_all_sorted(*kmer[k], *score[k], *row[k])) {
                    rest_h[R] = hp[k];
                    rest_k[R] = kmer[k];
                    rest_s[R] = score[k];
                    rest_r[R] = row[k];
                    R++;
                }
            if (R == 0) return;
            if (R < K) {
                pop_all_lockstep(R, rest_h, rest_k, rest_s, rest_r);
                return;
            }
        }
        pop_all_lockstep(K, hp, kmer, score, row);
    }
    static void pop_all_lockstep(int K, const BestHeap* const* hp, std::vector<uint64_t>* const* kmer, std::vector<double>* const* score,
                                 std::vector<uint64_t>* const* row) {
        bool ints = true;
        for (int k = 0; k < K; k++) ints = ints && hp[k]->ints_ok_;
        if (K == 1) {
            hp[0]->pop_all_classic(*kmer[0], *score[0], *row[0]);
            return;
        }
#define KGWAS_POP_CASE(N)                                                                       \
    case N:                                                                                     \
        if (ints) pop_all_multi<N, true>(hp, kmer, score, row); else pop_all_multi<N, false>(hp, kmer, score, row); \
        break;
        switch (K) {
            KGWAS_POP_CASE(2)
            KGWAS_POP_CASE(3)
            KGWAS_POP_CASE(4)
            KGWAS_POP_CASE(5)
            KGWAS_POP_CASE(6)
            KGWAS_POP_CASE(7)
            default: if (ints) pop_all_multi<8, true>(hp, kmer, score, row); else pop_all_multi<8, false>(hp, kmer, score, row);
        }
#undef KGWAS_POP_CASE
    }
    template <int K, bool INT = false>
    static void pop_all_multi(const BestHeap* const* hp, std::vector<uint64_t>* const* kmer, std::vector<double>* const* score,
                              std::vector<uint64_t>* const* row) {
        std::vector<Ent> tmp[K];
        Ent* a[K];
        const size_t n = hp[0]->v_.size();
        for (int k = 0; k < K; k++) {
            tmp[k].assign(hp[k]->v_.begin(), hp[k]->v_.end());
            a[k] = tmp[k].data();
            kmer[k]->resize(n);
            score[k]->resize(n);
            row[k]->resize(n);
        }
        for (size_t i = 0; i < n; i++) {
            const ptrdiff_t m = (ptrdiff_t)(n - i);
            Ent v[K];
            ptrdiff_t h[K], c[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                (*kmer[k])[i] = hp[k]->pay_[a[k][0].slot].kmer;
                (*score[k])[i] = a[k][0].score;
                (*row[k])[i] = hp[k]->pay_[a[k][0].slot].row;
                v[k] = a[k][m - 1];
                h[k] = 0;
                c[k] = 0;
            }
            if (m <= 1) continue;
            const ptrdiff_t len = m - 1, lim = (len - 1) / 2;
            for (;;) {
                bool any = false;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (c[k] < lim) {
                        ptrdiff_t cc = 2 * (c[k] + 1);
                        cc -= gt<INT>(a[k][cc], a[k][cc - 1]) ? 1 : 0;
                        a[k][h[k]] = a[k][cc];
                        h[k] = cc;
                        c[k] = cc;
                        any = true;
                    }
                }
                if (!any) break;
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
                    c[k] = 2 * (c[k] + 1);
                    a[k][h[k]] = a[k][c[k] - 1];
                    h[k] = c[k] - 1;
                }
                ptrdiff_t hh = h[k], p = (hh - 1) / 2;
                while (hh > 0 && gt<INT>(a[k][p], v[k])) {
                    a[k][hh] = a[k][p];
                    hh = p;
                    p = (hh - 1) / 2;
                }
                a[k][hh] = v[k];
            }
        }
    }

    // ---- eviction ring (record_history = 2) --------------------------------------------------------------------
    // What a later shard has to contribute to a cross-shard merge is its effective pushes with a score above the
    // earlier shards' final minimum T. Every effective push is either still in the heap or was evicted as the
    // minimum of its time, and evicted scores only rise: the pushes above T are the heap's entries above T plus
    // the LAST few evictions (T and this heap's own final minimum are N-th order statistics of equally large
    // shards: they differ by O(sqrt N) ranks). So instead of logging all N(1 + ln(M/N)) pushes, the last R
    // evictions are kept in a small cache-resident ring.
    struct Rec {
        uint64_t kmer;
        double score;
        uint64_t row;
    };
    void enable_ring(size_t r) {
        ring_.assign(r, Rec{0, 0, 0});
        evicted_ = 0;
    }
    inline bool ring_enabled() const { return !ring_.empty(); }
    // The effective pushes with score > thr (thr = -inf: all of them) in row order, appended to `out`. Returns false
    // if evictions that may have qualified have already left the ring.
    bool pushes_above(double thr, std::vector<Rec>& out) const {
        const bool all = thr == -__builtin_huge_val();
        const size_t R = ring_.size(), kept = evicted_ < R ? (size_t)evicted_ : R;
        if (evicted_ > R) {  // the oldest surviving eviction bounds everything that was dropped from below it
            const Rec& oldest = ring_[evicted_ % R];
            if (all || oldest.score > thr) return false;
        }
        const size_t first = out.size();
        out.reserve(first + v_.size() + kept);
        for (const Ent& e : v_)
            if (all || e.score > thr) out.push_back(Rec{pay_[e.slot].kmer, e.score, pay_[e.slot].row});
        for (size_t i = 0; i < kept; i++)
            if (all || ring_[i].score > thr) out.push_back(ring_[i]);
        sort_by_row(out.data() + first, out.size() - first);
        return true;
    }
    // Row order (rows are unique within a column). This runs once per column on the merge's critical path, on ~N records
    // whose rows span one shard: (row - smallest row, index) pairs packed into 64-bit keys are sorted by LSD radix passes
    // of 11 bits (2048 counters: they stay in the L1, where the 65 536 counters of a 16-bit pass cost more than the
    // records did), then the 24-byte records are gathered once.
    static void sort_by_row(Rec* a, size_t n) {
        if (n < 2) return;
        uint64_t lo = a[0].row, hi = a[0].row;
        for (size_t i = 1; i < n; i++) {
            lo = a[i].row < lo ? a[i].row : lo;
            hi = a[i].row > hi ? a[i].row : hi;
        }
        const uint64_t span = hi - lo;
        constexpr int IDX_BITS = 24, DIGIT = 11;
        if (n < 64 || n >= (1ull << IDX_BITS) || span >= (1ull << (64 - IDX_BITS))) {
            std::sort(a, a + n, [](const Rec& x, const Rec& y) { return x.row < y.row; });
            return;
        }
        static thread_local std::vector<uint64_t> key0, key1;
        static thread_local std::vector<Rec> tmp;
        key0.resize(n);
        key1.resize(n);
        for (size_t i = 0; i < n; i++) key0[i] = ((a[i].row - lo) << IDX_BITS) | (uint64_t)i;
        uint64_t* src = key0.data();
        uint64_t* dst = key1.data();
        uint32_t cnt[1u << DIGIT];
        for (int shift = IDX_BITS; (span >> (shift - IDX_BITS)) != 0; shift += DIGIT) {
            memset(cnt, 0, sizeof(cnt));
            for (size_t i = 0; i < n; i++) cnt[(src[i] >> shift) & ((1u << DIGIT) - 1u)]++;
            uint32_t run = 0;
            for (uint32_t& c : cnt) {
                const uint32_t t = c;
                c = run;
                run += t;
            }
            for (size_t i = 0; i < n; i++) dst[cnt[(src[i] >> shift) & ((1u << DIGIT) - 1u)]++] = src[i];
            std::swap(src, dst);
        }
        tmp.assign(a, a + n);
        for (size_t i = 0; i < n; i++) a[i] = tmp[src[i] & ((1ull << IDX_BITS) - 1ull)];
    }

   private:
    inline void note_eviction(uint32_t slot, double score) {
        if (!ring_.empty()) {
            ring_[evicted_ % ring_.size()] = Rec{pay_[slot].kmer, score, pay_[slot].row};
            evicted_++;
        }
    }
    size_t n_res_;
    std::pmr::vector<Ent> v_;
    std::pmr::vector<Pay> pay_;
    std::vector<Rec> ring_;
    uint64_t evicted_ = 0;
    uint64_t inserted_, pushes_;
    double lowest_;
    bool ints_ok_ = true;  // every score inserted so far is in +0 .. +inf: gt<true> decides as gt<false> does
};

}  // namespace kgwas
