// scan_lazy.cpp — columns whose top-N is SELECTED instead of replayed.
//
// add_association (src/best_associations_heap.cpp:43-59) keeps the N largest scores; WHICH of several equal scores stay, and
// the order equal scores pop in (output_to_file_with_scores, :82-92), depend on libstdc++'s heap layout, i.e. on the whole
// history of effective pushes - that is why scan_replay.cpp replays every one of them (27 ns each, 10 M per 100 M-row pass
// at 101 columns: the scan's critical path on 16 CPUs). But where a column's N largest scores - and the (N + 1)-th - are
// pairwise DISTINCT, none of them NaN or negative, the result does not depend on the layout at all: the heap ends up
// holding exactly the N largest, and N pops from ANY valid heap of distinct keys yield them in ascending order. Such a
// column needs a selection, not a replay.
//
// A lazy column keeps
//   * its LOG: every record the column was shipped (the dense chunks' MAC-passing rows, then the sparse chunks' candidates)
//     in row order; what an exact replay would need, should it come to that. A sparse chunk's records are not copied: they
//     stay where the GPU wrote them, in the session's pinned record ring, and the log holds a reference (LazyCol::Seg) - the
//     ring is filled linearly and not recycled while logs refer to it (scan_internal.h, ring_keep; sized for a pass's records:
//     390 MB at 101 columns, 775 MB at 201). If it runs full after all, the logs take copies (detach) and the ring recycles;
//     from then on every chunk's records are copied into the logs (three memcpys), as all of them were before.
//   * its POOL: (score bits, k-mer, row) of the records that can still be among the N largest. It is DERIVED from the log:
//     every ~4 N logged records - and whenever somebody needs the pool - the records not looked at yet are scanned, scores
//     only, and those at or above the column's bound as it stands THEN join the pool (a record that would have passed when
//     it arrived and no longer does is never touched: 1 ns per record instead of the 7.5 ns of copy + compare + insert). The
//     bound is the device's own threshold behind the latest chunk (a valid lower bound of the reference heap's minimum, within
//     0.4 % of the N-th largest score). When the pool holds 2 N entries it is PRUNED with the bound (one linear pass);
//     should that leave more than 2 N, nth_element finds the N-th largest score v, everything below v goes, entries EQUAL to v
//     stay (a tie at the boundary must remain visible), and v becomes the bound.
// At finish the pool is cut to its N largest and sorted; if more than N entries are >= the N-th score (boundary tie), two
// neighbours are equal (tie inside), or the column ever saw a NaN / negative score, the column is MATERIALISED instead: its log
// is replayed through its BestHeap, exactly as the streaming replay would have done, and popped. Tables whose rows repeat
// presence/absence patterns (real k-mer tables: the rule) tie all over the place - a column whose first dense chunk shows a
// tie among its top N is materialised on the spot and replayed from then on, as every column was before round 5; the pools
// are looked at again for ties behind the chunks that cross 40 and 65 % of a large feed (a radix sort of ~N keys), so that a
// column that will need the replay gets it beside the GPU's work; a tie that turns up later still is replayed - ~4 ms at
// N = 10 001 and 10^8 rows, all such columns side by side - by the replay workers before they leave the last feed
// (scan_replay.cpp, drain) or, without kgwas_scan_expect_finish, by kgwas_scan_finish.
// KGWAS_FULL_REPLAY=1: no column is ever lazy (the effective-push count, which the tests use as a tripwire, exists for
// replayed columns only).
#include "scan_internal.h"

namespace kgwas {

void LazyCol::reset(bool enable, uint64_t topn_) {
    on = enable;
    bad = false;
    topn = topn_;
    n_logged = 0;
    bound_bits = 0;
    have_bound = false;
    n_prunes = 0;
    l_n = 0;
    segs.clear();
    scan_seg = 0, scan_off = 0, unscanned = 0;
    pool.clear();
    if (enable && pool.capacity() < 3 * (size_t)std::min<uint64_t>(topn_, 1u << 22) + 64) pool.reserve(3 * (size_t)std::min<uint64_t>(topn_, 1u << 22) + 64);
}

void LazyCol::reserve_log(size_t need) {
    if (need <= l_cap) return;
    size_t nc = l_cap ? l_cap + l_cap / 2 : (size_t)1 << 14;
    if (nc < need) nc = need;
    void* a = realloc(l_sc, nc * sizeof(double));
    if (!a) throw std::bad_alloc();
    l_sc = static_cast<double*>(a);
    void* b = realloc(l_km, nc * sizeof(uint64_t));
    if (!b) throw std::bad_alloc();
    l_km = static_cast<uint64_t*>(b);
    void* c = realloc(l_rw, nc * sizeof(uint64_t));
    if (!c) throw std::bad_alloc();
    l_rw = static_cast<uint64_t*>(c);
    l_cap = nc;
}

// A sparse chunk's records of this column: into the log as they are - by reference, or three copies (a narrow chunk's
// placeholders - survivors that are no candidates, score -inf - travel along and are skipped by every reader); the device's
// threshold behind this very chunk raises the bound. The pool sees them at the next scan_pending.
void LazyCol::take_chunk(const double* sc, const uint64_t* km, const uint32_t* rw, uint32_t n, uint64_t row0, uint64_t thr_bits, bool by_ref) {
    // (a threshold of 0 is "none yet": it must not pass for a bound that a selection has established - lazy_lowest)
    if (thr_bits != 0 && thr_bits <= 0x7FF0000000000000ull && (!have_bound || thr_bits > bound_bits)) {
        bound_bits = thr_bits;
        have_bound = true;
    }
    if (n == 0) return;
    if (by_ref) {
        segs.push_back(Seg{sc, km, rw, row0, 0, n});
    } else {
        reserve_log(l_n + n);
        memcpy(l_sc + l_n, sc, (size_t)n * sizeof(double));
        memcpy(l_km + l_n, km, (size_t)n * sizeof(uint64_t));
        uint64_t* dr = l_rw + l_n;
        for (uint32_t i = 0; i < n; i++) dr[i] = row0 + rw[i];
        l_n += n;
        owned_appended(n);
    }
    n_logged += n;
    unscanned += n;
    if (unscanned >= 4 * topn + 1024) scan_pending();
}

void LazyCol::scan_pending() {
    if (!unscanned) return;
    uint64_t lim = have_bound ? bound_bits : 0;
    auto look = [&](uint64_t b, uint64_t km, uint64_t rw) {
        if (b > 0x7FF0000000000000ull) {  // NaN, or the sign bit set
            if (b != NEG_INF)
                bad = true;
            else
                n_logged--;  // -inf: a narrow chunk's survivor that was no candidate - no add_association call, no record (full())
            return;
        }
        pool.push_back(Ent{b, km, rw});
        if (pool.size() >= 2 * (size_t)topn + 64) {
            prune(bound_bits);
            lim = have_bound ? bound_bits : 0;  // (a selection may have raised it)
        }
    };
    for (; scan_seg < segs.size(); scan_seg++, scan_off = 0) {
        const Seg& g = segs[scan_seg];
        if (g.sc) {
            const uint64_t* sb = reinterpret_cast<const uint64_t*>(g.sc);
            for (uint32_t i = scan_off; i < g.n; i++)
                if (sb[i] >= lim) look(sb[i], g.km[i], g.row0 + g.rw[i]);
        } else {
            const uint64_t* sb = reinterpret_cast<const uint64_t*>(l_sc);
            for (size_t i = g.off + scan_off; i < g.off + g.n; i++)
                if (sb[i] >= lim) look(sb[i], l_km[i], l_rw[i]);
        }
    }
    // (the last segment may still grow - single records are appended to an owned one: it is looked at again from where this
    // pass ended)
    if (!segs.empty()) {
        scan_seg = segs.size() - 1;
        scan_off = segs.back().n;
    }
    unscanned = 0;
}

// Every referenced segment becomes an owned copy (the order of the log is kept): one pass, new arrays.
void LazyCol::detach() {
    scan_pending();
    bool any = false;
    size_t total = 0;
    for (const Seg& g : segs) {
        any |= g.sc != nullptr;
        total += g.n;
    }
    if (!any) return;
    if (!total) {  // only empty references: nothing to copy, nothing left to refer to
        segs.clear();
        scan_seg = 0, scan_off = 0;
        return;
    }
    double* nsc = static_cast<double*>(malloc(std::max<size_t>(total, 1) * sizeof(double)));
    uint64_t* nkm = static_cast<uint64_t*>(malloc(std::max<size_t>(total, 1) * sizeof(uint64_t)));
    uint64_t* nrw = static_cast<uint64_t*>(malloc(std::max<size_t>(total, 1) * sizeof(uint64_t)));
    if (!nsc || !nkm || !nrw) {
        free(nsc), free(nkm), free(nrw);
        throw std::bad_alloc();
    }
    size_t o = 0;
    for_each([&](double sc, uint64_t km, uint64_t rw) {
        nsc[o] = sc;
        nkm[o] = km;
        nrw[o] = rw;
        o++;
    });
    free(l_sc), free(l_km), free(l_rw);
    l_sc = nsc, l_km = nkm, l_rw = nrw;
    l_n = l_cap = total;
    segs.clear();
    size_t at = 0;
    while (at < total) {  // (a segment counts its records in 32 bits)
        const size_t c = std::min<size_t>(total - at, 0xFFFFFFFFull);
        segs.push_back(Seg{nullptr, nullptr, nullptr, 0, at, (uint32_t)c});
        at += c;
    }
    scan_seg = segs.size() - 1;
    scan_off = segs.back().n;
}

// thr: a valid lower bound of the reference heap's minimum (the device's threshold behind some chunk: the highest boundary of
// its score histogram with at least N counted scores at or above it, within 0.4 % of the N-th largest). Everything below it
// goes - one linear pass instead of a selection; entries EQUAL to it stay (it may be the N-th largest itself).
void LazyCol::prune(uint64_t thr_bits) {
    n_prunes++;
    if (thr_bits <= 0x7FF0000000000000ull && (!have_bound || thr_bits > bound_bits)) {  // (NaN: a frozen column)
        bound_bits = thr_bits;
        have_bound = true;
    }
    if (have_bound) {  // (entries that came in under an earlier, lower bound)
        size_t keep = 0;
        for (size_t i = 0; i < pool.size(); i++)
            if (pool[i].bits >= bound_bits) pool[keep++] = pool[i];
        pool.resize(keep);
    }
    if (pool.size() >= 2 * (size_t)topn + 64) compact();  // (a threshold that lags far behind: select after all)
}

// the pool's N largest stay (with everything equal to the N-th); the N-th becomes the bound
void LazyCol::compact() {
    const size_t N = (size_t)topn;
    if (pool.size() <= N) return;
    std::nth_element(pool.begin(), pool.begin() + (N - 1), pool.end(), [](const Ent& a, const Ent& b) { return a.bits > b.bits; });
    const uint64_t v = pool[N - 1].bits;
    size_t keep = N;
    for (size_t i = N; i < pool.size(); i++)
        if (pool[i].bits == v) pool[keep++] = pool[i];
    pool.resize(keep);
    bound_bits = v;
    have_bound = true;
}

// Result lists by selection (any number of entries: a heap that never filled holds them all). false: the scores alone do not
// decide the result - the caller materialises the column.
bool LazyCol::select(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row) {
    scan_pending();
    if (bad) return false;
    const size_t N = (size_t)topn;
    // the pool down to ~1.04 N entries with one linear pass (the bound is within 0.4 % of the N-th largest score); a selection
    // only if a lagging bound leaves half as many again
    prune(bound_bits);
    if (pool.size() > N + N / 2) compact();
    const size_t n = pool.size();
    // ascending by bits: LSD radix over the digits in which the keys differ (as BestHeap::pop_all_sorted), on (key, index) pairs
    struct KI {
        uint64_t bits;
        uint32_t idx;
    };
    static thread_local std::vector<KI> a, b;
    if (a.size() < n) a.resize(n), b.resize(n);
    uint64_t all_or = 0, all_and = ~0ull;
    for (size_t i = 0; i < n; i++) {
        a[i] = KI{pool[i].bits, (uint32_t)i};
        all_or |= pool[i].bits;
        all_and &= pool[i].bits;
    }
    const uint64_t varying = all_or ^ all_and;
    KI* src = a.data();
    KI* dst = b.data();
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
    // the N largest are src[lo .. n); an entry below them that equals the N-th largest: which of the equals stay is the layout's
    // business, and so is the pop order of two equals among the N
    const size_t lo = n > N ? n - N : 0;
    for (size_t i = lo ? lo : 1; i < n; i++)
        if (src[i].bits == src[i - 1].bits) return false;
    const size_t m = n - lo;
    kmer.resize(m);
    score.resize(m);
    row.resize(m);
    for (size_t i = 0; i < m; i++) {
        const Ent& e = pool[src[lo + i].idx];
        kmer[i] = e.kmer;
        memcpy(&score[i], &e.bits, 8);
        row[i] = e.row;
    }
    return true;
}

// Are two of the pool's N largest scores equal, or does the (N + 1)-th equal the N-th, right now? The pool is pruned with its
// bound (~1.04 N entries stay), its keys are radix-sorted (LSD over the digits in which they differ: the N largest scores of a
// column share sign, exponent and often the leading mantissa bits) and the top N + 1 compared with their neighbours: ~80 us
// at N = 10 001.
bool LazyCol::ties_now() {
    scan_pending();
    if (bad) return true;
    const size_t N = (size_t)topn;
    prune(bound_bits);
    if (pool.size() > N + N / 2) compact();
    const size_t n = pool.size();
    static thread_local std::vector<uint64_t> k, k2;
    if (k.size() < n) k.resize(n), k2.resize(n);
    uint64_t all_or = 0, all_and = ~0ull;
    for (size_t i = 0; i < n; i++) {
        k[i] = pool[i].bits;
        all_or |= k[i];
        all_and &= k[i];
    }
    const uint64_t varying = all_or ^ all_and;
    uint64_t* src = k.data();
    uint64_t* dst = k2.data();
    for (int shift = 0; shift < 64; shift += 11) {
        if (((varying >> shift) & 0x7FFull) == 0) continue;
        uint32_t cnt[2048] = {0};
        for (size_t i = 0; i < n; i++) cnt[(src[i] >> shift) & 0x7FFu]++;
        uint32_t run = 0;
        for (int d = 0; d < 2048; d++) {
            const uint32_t c = cnt[d];
            cnt[d] = run;
            run += c;
        }
        for (size_t i = 0; i < n; i++) dst[cnt[(src[i] >> shift) & 0x7FFu]++] = src[i];
        std::swap(src, dst);
    }
    // ascending: the N largest are src[n - N .. n), the (N + 1)-th is src[n - N - 1]
    const size_t lo = n > N ? n - N - 1 : 0;
    for (size_t i = lo + 1; i < n; i++)
        if (src[i] == src[i - 1]) return true;
    return false;
}

// The column leaves select mode: its log goes through its (empty) heap, in row order, as the streaming replay would have
// sent it; returns the effective pushes.
uint64_t lazy_materialize(kgwas_scan* s, size_t j) {
    LazyCol& L = s->lazy[j];
    if (!L.on) return 0;
    BestHeap& h = s->heaps[j];
    uint64_t pushes = 0;
    const double none = -std::numeric_limits<double>::infinity();
    L.for_each([&](double sc, uint64_t km, uint64_t rw) {
        if (sc == none) return;  // (a narrow chunk's survivor that was no candidate)
        if (h.add(km, sc, (size_t)rw)) pushes++;
    });
    L.on = false;
    L.release_log();
    std::vector<LazyCol::Ent>().swap(L.pool);
    return pushes;
}

// Every column starts a scan in select mode (or none does: KGWAS_FULL_REPLAY=1, sessions that record push histories for a
// cross-shard merge, exact-scorer sessions).
void lazy_reset(kgwas_scan* s) {
    s->lazy.resize(s->n_pheno);
    for (uint64_t j = 0; j < s->n_pheno; j++) s->lazy[j].reset(s->lazy_enabled, s->topn[j]);
    s->lazy_any.store(s->lazy_enabled, std::memory_order_relaxed);
    s->n_selected.store(0);
    s->n_unselected.store(0);
    s->lazy_pushes.store(0);
    s->tie_check_rows = 4ull << 20;
    // nothing refers to the record ring any more; a session in select mode fills it linearly (scan_internal.h, ring_keep)
    s->ring_head = s->ring_tail = 0;
    s->ring_keep.store(s->lazy_enabled && s->coarse && !(opt_int("KGWAS_LOG_BY_REF", 1) == 0), std::memory_order_release);
}

// BestHeap::lowest() / full() of a column in select mode without giving it a heap: a full heap's minimum is the N-th largest
// score it was offered (whatever the ties), one that is still filling reports the smallest. false: the column saw a NaN or
// negative score - only its heap knows (the caller materialises it).
bool lazy_lowest(kgwas_scan* s, size_t j, double* lowest, bool* full) {
    LazyCol& L = s->lazy[j];
    L.scan_pending();
    if (L.bad) return false;
    *full = L.full();
    if (L.full()) {
        // The heap's minimum is the N-th largest score logged. More than N entries in the pool: the selection finds it (and
        // makes it the bound). N or fewer: the pool holds exactly the records at or above the bound, and since N records were
        // logged and at most N passed, the N-th largest is the pool's smallest entry - NOT the bound, which is a device
        // threshold and only a lower bound of it (up to 0.4 % low).
        uint64_t v = ~0ull;
        if (L.pool.size() > (size_t)L.topn) {
            L.compact();
            v = L.bound_bits;
        } else {
            for (const LazyCol::Ent& e : L.pool) v = std::min(v, e.bits);
            if (L.pool.empty()) v = L.have_bound ? L.bound_bits : 0;  // (unreachable while the bound is a lower bound; kept total)
        }
        memcpy(lowest, &v, 8);
    } else {
        uint64_t v = ~0ull;
        for (const LazyCol::Ent& e : L.pool) v = std::min(v, e.bits);
        if (L.pool.empty()) v = 0;  // (BestHeap: lowest_ = 0 until something is pushed)
        memcpy(lowest, &v, 8);
    }
    return true;
}

void lazy_materialize_all(kgwas_scan* s) {
    if (!s->lazy_any.load(std::memory_order_relaxed)) return;
    std::atomic<uint64_t> pushes(0);
    s->pool->parallel_for(s->n_pheno, [&](size_t j) { pushes += lazy_materialize(s, j); });
    s->st.heap_pushes += pushes.load();
    s->lazy_any.store(false, std::memory_order_relaxed);
    s->ring_keep.store(false, std::memory_order_release);  // (no log is left: the next feed recycles the ring)
    refresh_full(s);
}

// Result lists of column j (finish, or an idle worker at the tail of the last feed): by selection where the scores decide,
// else through the heap.
void lazy_finish_column(kgwas_scan* s, size_t j, bool known_tie) {
    LazyCol& L = s->lazy[j];
    if (L.on) {
        static const bool trace = exp_set("KGWAS_FINISH_TRACE");
        const double t0 = trace ? s->t_ms() : 0;
        if (!known_tie && L.select(s->res_kmer[j], s->res_score[j], s->res_row[j])) {
            s->n_selected.fetch_add(1, std::memory_order_relaxed);
            if (trace) fprintf(stderr, "finish col %zu: select %.3f ms\n", j, s->t_ms() - t0);
            return;
        }
        const double t1 = trace ? s->t_ms() : 0;
        const size_t ln = (size_t)L.n_logged;
        const uint64_t pu = lazy_materialize(s, j);
        s->lazy_pushes.fetch_add(pu, std::memory_order_relaxed);
        s->n_unselected.fetch_add(1, std::memory_order_relaxed);
        const double t2 = trace ? s->t_ms() : 0;
        s->heaps[j].pop_all(s->res_kmer[j], s->res_score[j], s->res_row[j]);
        if (trace) fprintf(stderr, "finish col %zu: failed select %.3f, materialize %.3f (%zu records, %llu pushes), pops %.3f ms\n", j, t1 - t0, t2 - t1, ln, (unsigned long long)pu, s->t_ms() - t2);
        return;
    }
    s->heaps[j].pop_all(s->res_kmer[j], s->res_score[j], s->res_row[j]);
}

}  // namespace kgwas

// The select-or-replay decision of one column on the host alone (include/kgwas.h, kgwas_select_check): the CPU suite feeds it
// record streams with and without ties and holds the lists against the heap mirror's.
extern "C" int kgwas_select_check(uint64_t topn, uint64_t n_chunks, const uint64_t* chunk_n, const uint64_t* chunk_row0, const double* chunk_thr,
                                  const double* score, const uint64_t* kmer, const uint32_t* row_in_chunk, int by_ref,
                                  int64_t detach_after_chunk, int* selected, uint64_t* out_kmer, double* out_score, uint64_t* out_row,
                                  uint64_t* out_n) {
    using namespace kgwas;
    return guarded([&] {
        if (!topn || !chunk_n || !selected || !out_n) throw Error(KGWAS_ERR_ARG, "kgwas_select_check: bad argument");
        LazyCol L;
        L.reset(true, topn);
        size_t at = 0;
        for (uint64_t c = 0; c < n_chunks; c++) {
            const uint32_t n = (uint32_t)chunk_n[c];
            uint64_t tb;
            memcpy(&tb, &chunk_thr[c], 8);
            L.take_chunk(score + at, kmer + at, row_in_chunk + at, n, chunk_row0[c], tb, by_ref != 0);
            at += n;
            if ((int64_t)c == detach_after_chunk) L.detach();
        }
        std::vector<uint64_t> km, rw;
        std::vector<double> sc;
        if (L.select(km, sc, rw)) {
            *selected = 1;
        } else {
            *selected = 0;
            BestHeap h((size_t)topn);
            const double none = -std::numeric_limits<double>::infinity();
            L.for_each([&](double s_, uint64_t k_, uint64_t r_) {
                if (s_ != none) h.add(k_, s_, (size_t)r_);
            });
            h.pop_all(km, sc, rw);
        }
        *out_n = km.size();
        for (size_t i = 0; i < km.size(); i++) {
            if (out_kmer) out_kmer[i] = km[i];
            if (out_score) out_score[i] = sc[i];
            if (out_row) out_row[i] = rw[i];
        }
    });
}
