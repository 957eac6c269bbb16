// scan_internal.h — the association-scan session's internals shared by scan_host.cpp (CPU topology, heaps), scan_gpu.cpp
// (chunk submission: kernels, thresholds, record copies), scan_replay.cpp (streaming replay + the feed loop),
// scan_create.cpp (session set-up: operand sets of the filters) and scan_api.cpp (the C ABI). Not a public header: the
// boundary is include/kgwas.h.
//
// The association-scan session: pass 1 of associate_kmers (src/associate_kmers.cpp:99-148)
// re-designed around the GPU.
//
//   reference                                    here
//   ---------------------------------------     -------------------------------------------------
//   load_kmers: read, MAC filter, squeeze        rows stream from HBM in file layout; MAC predicate
//   (serial, per-bit)                            and (only if the column map is not the identity
//                                                prefix) a squeeze kernel, per device chunk
//   one CTPL task per phenotype column           one kernel scores every (k-mer, column) pair of a
//   scoring the batch (SSE) into its heap        chunk; only pairs that beat a stale heap minimum
//                                                come back; the host replays them, in row order,
//                                                through the same std::priority_queue
//
// Exactness argument (SURVEY.md §7 hard part 1): once a heap is full add_association is a
// no-op unless score > lowest_score, and lowest_score never decreases. A row whose score is
// <= ANY earlier value of lowest_score can therefore be dropped without changing the heap's
// history. Until every heap is full the chunks run in dense mode (all scores come back).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>

#include "common.h"
#include "heap.h"
#include "ingest.h"
#include "kernels.h"

namespace kgwas {

// Minimal persistent worker pool: parallel_for over phenotype columns. The caller does not take part: it
// sleeps until the workers are done, so exactly size() threads are busy (sized to the CPU quota).
class Pool {
   public:
    // cpus: optional CPU to pin worker i to (empty = leave placement to the scheduler).
    explicit Pool(unsigned n, const std::vector<std::vector<int>>& cpus = {})
        : stop_(false), gen_(0), n_items_(0) {
        if (n < 1) n = 1;
        for (unsigned i = 0; i < n; i++) {
            const std::vector<int> mine = i < cpus.size() ? cpus[i] : std::vector<int>();
            try {
                th_.emplace_back([this, i, mine] {
                    kgwas_name_this_thread("kgwas-replay");
                    if (!mine.empty()) {
                        cpu_set_t set;
                        CPU_ZERO(&set);
                        for (int c : mine)
                            if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
                        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
                    }
                    loop(i);
                });
            } catch (const std::system_error&) {
                // no more threads to be had: the pool is the workers that exist (items are claimed, not assigned, so any
                // number of workers runs them all); with none at all the session cannot be built
                if (th_.empty()) throw;
                break;
            }
        }
    }
    ~Pool() {
        {
            std::unique_lock<std::mutex> lk(mu_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // Static assignment: item i always runs on worker i % size(), so a phenotype column's heap
    // (~320 KB at N = 10001) stays in one core's cache from chunk to chunk.
    void parallel_for(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        start(n, fn);
        wait();
        rethrow();
    }
    // An item that throws (an allocation that fails) ends its worker's turn; the first such exception is kept until the next
    // start() and rethrown here: by parallel_for itself, and by callers of start()/wait() once they are on their normal path.
    void rethrow() {
        std::exception_ptr e;
        {
            std::unique_lock<std::mutex> lk(mu_);
            e = err_;
            err_ = nullptr;
        }
        if (e) std::rethrow_exception(e);
    }
    // The two halves of parallel_for: start() hands the items out and returns, wait() blocks until every ITEM is done - not
    // until every worker has looked in: on a shared host a worker that was asleep may get its CPU milliseconds late, and its
    // items are long done by the others (they are claimed, not assigned) - the dense fill of a headline step, 1.4 ms of work,
    // took 6-9 ms in one step of twenty on such boxes while wait() counted workers. Between the two the workers run on their own
    // (the streaming replay: items are worker loops that end when told to). fn must stay alive until wait() returns; one start
    // at a time.
    void start(size_t n, const std::function<void(size_t)>& fn) {
        std::unique_lock<std::mutex> lk(mu_);
        // (a worker of the previous start that woke late may still be walking the claim flags - it finds nothing - before they
        // are reset: workers enter run() under mu_, so none slips in behind this check)
        while (in_run_.load(std::memory_order_acquire) != 0) {
            lk.unlock();
            __builtin_ia32_pause();
            lk.lock();
        }
        fn_ = &fn;
        n_items_ = n;
        if (claimed_.size() < n) claimed_ = std::vector<std::atomic<uint8_t>>(n);
        for (size_t i = 0; i < n; i++) claimed_[i].store(0, std::memory_order_relaxed);
        left_.store(n, std::memory_order_relaxed);
        err_ = nullptr;
        done_.store(n == 0 ? 1 : 0, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        lk.unlock();
        cv_.notify_all();
    }
    void wait(bool spin = true) {
        for (int sp = 0; spin && sp < 20000; sp++) {  // replays take a few ms at most: spin before sleeping
            if (done_.load(std::memory_order_acquire)) break;
            __builtin_ia32_pause();
        }
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return done_.load(std::memory_order_acquire) != 0; });
    }
    size_t size() const { return th_.size(); }
    bool finished() const { return done_.load(std::memory_order_acquire) != 0; }  // every started item has run

   private:
    // Own items first, in increasing order; then take whatever nobody has started yet, from the far end
    // (with 101 columns on 16 workers the five workers that own a seventh column give it away to a worker
    // that is done with its six). An item runs exactly once, on one thread; one that throws (an allocation that fails) is
    // done all the same, and the first such exception is kept for rethrow().
    void item(size_t i) {
        try {
            (*fn_)(i);
        } catch (...) {
            std::unique_lock<std::mutex> lk(mu_);
            if (!err_) err_ = std::current_exception();
        }
        if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::unique_lock<std::mutex> lk(mu_);
            done_.store(1, std::memory_order_release);
            done_cv_.notify_all();
        }
    }
    void run(size_t me) {
        const size_t T = th_.size();
        for (size_t i = me; i < n_items_; i += T)
            if (!claimed_[i].exchange(1, std::memory_order_acq_rel)) item(i);
        for (size_t i = n_items_; i-- > 0;)
            if (!claimed_[i].load(std::memory_order_relaxed) && !claimed_[i].exchange(1, std::memory_order_acq_rel)) item(i);
    }
    void loop(size_t me) {
        uint64_t seen = 0;
        for (;;) {
            // Chunks arrive every few milliseconds while a scan is running: spin briefly before
            // sleeping so the wake-up does not cost a futex round trip per worker per chunk.
            bool got = false;
            for (int spin = 0; spin < 4000; spin++) {
                if (gen_.load(std::memory_order_acquire) != seen) {
                    got = true;
                    break;
                }
                __builtin_ia32_pause();
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (!got) cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                seen = gen_.load(std::memory_order_acquire);
                if (stop_) return;
                in_run_.fetch_add(1, std::memory_order_acq_rel);  // (under mu_: start() looks at it there)
            }
            run(me);
            in_run_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    bool stop_;
    std::atomic<uint64_t> gen_;
    std::atomic<int> done_{0};
    std::atomic<int> in_run_{0};      // workers between the claim of their first item and their last look at the flags
    std::atomic<size_t> left_{0};     // items of the current start that have not finished
    std::vector<std::atomic<uint8_t>> claimed_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_items_;
    std::exception_ptr err_;  // the first exception an item of the current start() threw (guarded by mu_)
};

// scan_host.cpp: one CPU per replay worker near the GPU; the CPUs the process may really use (cgroup quota)
std::vector<std::vector<int>> pick_replay_cpus(unsigned n, int device);
unsigned usable_cpus();

// Effective pushes of one phenotype column in row order (record_history: what a later shard contributes to the
// cross-shard merge). An append-only log of 24-byte records written with streaming stores: 10 M records per pass go
// straight to memory instead of through the worker's L2, where they would evict the heaps the same thread is
// updating (recording through three std::vectors cost 12 ms per 36 ms pass). The separate arrays that
// kgwas_scan_history hands out are made on demand.
struct History {
    struct Rec {
        uint64_t kmer;
        double score;
        uint64_t row;
    };
    Rec* p = nullptr;
    size_t n = 0, cap = 0;
    std::vector<uint64_t> v_kmer, v_row;  // kgwas_scan_history's views
    std::vector<double> v_score;
    History() = default;
    History(const History&) = delete;
    History& operator=(const History&) = delete;
    History(History&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    ~History() { free(p); }
    inline void push(uint64_t kmer, double score, uint64_t row) {
        if (n == cap) grow();
        p[n] = Rec{kmer, score, row};
        n++;
    }
    void grow() {
        const size_t nc = cap ? cap * 2 : (1u << 14);
        void* q = nullptr;
        if (posix_memalign(&q, 64, nc * sizeof(Rec)) != 0) throw std::bad_alloc();
        _mm_sfence();  // our own streaming stores must have landed before they are copied
        if (n) memcpy(q, p, n * sizeof(Rec));
        free(p);
        p = static_cast<Rec*>(q);
        cap = nc;
    }
    void clear() { n = 0; }
    void release() {
        free(p);
        p = nullptr;
        n = cap = 0;
    }
};

// A column whose top-N is selected instead of replayed (scan_lazy.cpp).
struct LazyCol {
    struct Ent {
        uint64_t bits;  // the score's bit pattern (+0 .. +inf: ordered like the scores)
        uint64_t kmer, row;
    };
    static constexpr uint64_t NEG_INF = 0xFFF0000000000000ull;  // "a survivor that is no candidate" in a narrow chunk's records
    bool on = false;   // the column is in select mode (its BestHeap is empty)
    bool bad = false;  // a NaN or negative score came by: the heap's order is not the scores' order - materialise at finish
    uint64_t topn = 0;
    uint64_t n_logged = 0;  // records logged (dense chunks: every MAC-passing row - "is the heap full")
    uint64_t bound_bits = 0;
    bool have_bound = false;
    uint32_t n_prunes = 0;
    // the LOG: segments in row order. A sparse chunk's records stay where the GPU put them - in the session's pinned record
    // ring, which is not recycled while columns refer to it (fetch_records) - and the log holds a reference (sc != null);
    // single records (dense chunks' rows, absorbed shards, exact-scorer candidates) and, once the ring has run full and gone
    // back to recycling, the chunks' records as well are copied into the three owned arrays (sc == null: [off, off + n) there).
    struct Seg {
        const double* sc;
        const uint64_t* km;
        const uint32_t* rw;  // rows within the chunk
        uint64_t row0;
        size_t off;
        uint32_t n;
    };
    std::vector<Seg> segs;
    double* l_sc = nullptr;
    uint64_t* l_km = nullptr;
    uint64_t* l_rw = nullptr;
    size_t l_n = 0, l_cap = 0;  // owned records
    std::vector<Ent> pool;
    // every logged record, in row order: f(score, kmer, row)
    template <class F>
    inline void for_each(F&& f) const {
        for (const Seg& g : segs) {
            if (g.sc) {
                for (uint32_t i = 0; i < g.n; i++) f(g.sc[i], g.km[i], g.row0 + g.rw[i]);
            } else {
                for (size_t i = g.off; i < g.off + g.n; i++) f(l_sc[i], l_km[i], l_rw[i]);
            }
        }
    }
    inline void owned_appended(size_t n) {  // n records were appended to the owned arrays at l_n - n
        if (!segs.empty() && !segs.back().sc && segs.back().off + segs.back().n == l_n - n && (uint64_t)segs.back().n + n < (1ull << 32))
            segs.back().n += (uint32_t)n;
        else
            segs.push_back(Seg{nullptr, nullptr, nullptr, 0, l_n - n, (uint32_t)n});
    }
    void detach();  // referenced records -> owned copies (the ring is about to be recycled)
    LazyCol() = default;
    LazyCol(const LazyCol&) = delete;
    LazyCol& operator=(const LazyCol&) = delete;
    LazyCol(LazyCol&& o) noexcept { *this = std::move(o); }
    LazyCol& operator=(LazyCol&& o) noexcept {
        if (this != &o) {
            release_log();
            on = o.on, bad = o.bad, topn = o.topn, n_logged = o.n_logged, bound_bits = o.bound_bits, have_bound = o.have_bound, n_prunes = o.n_prunes;
            l_sc = o.l_sc, l_km = o.l_km, l_rw = o.l_rw, l_n = o.l_n, l_cap = o.l_cap;
            o.l_sc = nullptr, o.l_km = nullptr, o.l_rw = nullptr, o.l_n = o.l_cap = 0;
            segs = std::move(o.segs);
            o.segs.clear();
            scan_seg = o.scan_seg, scan_off = o.scan_off, unscanned = o.unscanned;
            pool = std::move(o.pool);
        }
        return *this;
    }
    ~LazyCol() { release_log(); }
    void release_log() {
        free(l_sc);
        free(l_km);
        free(l_rw);
        l_sc = nullptr, l_km = nullptr, l_rw = nullptr;
        l_n = l_cap = 0;
        segs.clear();
        scan_seg = 0, scan_off = 0, unscanned = 0;
    }
    void reserve_log(size_t need);
    void reset(bool enable, uint64_t topn_);
    void compact();
    void prune(uint64_t thr_bits);
    bool select(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row);
    bool ties_now();
    // a sparse chunk's records of this column (row order; rw: rows within the chunk), the device's threshold behind the chunk
    // by_ref: the records stay where they are (see Seg)
    void take_chunk(const double* sc, const uint64_t* km, const uint32_t* rw, uint32_t n, uint64_t row0, uint64_t thr_bits, bool by_ref);
    inline bool full() const { return n_logged >= topn; }
    // a valid lower bound of what the reference heap's minimum is by now (0: none yet)
    inline double bound() const {
        double v = 0.0;
        if (have_bound) memcpy(&v, &bound_bits, 8);
        return v;
    }
    // one record (dense chunks' rows, records absorbed from other shards, exact-scorer candidates)
    inline void add(uint64_t kmer, double score, uint64_t row) {
        uint64_t b;
        memcpy(&b, &score, 8);
        if (b == NEG_INF) return;  // never an add_association call
        if (l_n == l_cap) reserve_log(l_n + 1);
        l_sc[l_n] = score;
        l_km[l_n] = kmer;
        l_rw[l_n] = row;
        l_n++;
        owned_appended(1);
        n_logged++;
        if (++unscanned >= 4 * topn + 1024) scan_pending();
    }
    // The POOL is derived from the log: the records that are not in it yet are looked at - one pass over their scores - and
    // those at or above the bound AS IT STANDS NOW join it (a record that would have passed when it arrived and does not pass
    // now is never touched again: looking late is looking at less). Called every ~4 N logged records and by everything that
    // reads the pool, `bad` or the N-th largest score.
    void scan_pending();
    size_t scan_seg = 0;   // segs[scan_seg] from record scan_off on, and every later segment, are not in the pool yet
    uint32_t scan_off = 0;
    uint64_t unscanned = 0;
};

// Evictions a heap of N entries keeps for the cross-shard merge (record_history = 2): the entries of a shard above
// another equally large shard's N-th score number N +- sqrt(2N); 16 of those deviations, unless KGWAS_HISTORY_RING
// says otherwise (`forced` > 1).
inline size_t ring_size(size_t forced, uint64_t topn) {
    if (forced > 1) return forced;
    const double r = 16.0 * std::sqrt(2.0 * (double)topn);
    return (size_t)std::min<double>(std::max<double>(r, 256.0), 1048576.0);
}

constexpr int MAX_SLOTS = 48;  // upper bound on sparse chunks the GPU may run ahead of the host replay

// What a worker adds up while replaying (chunk, column group) units.
struct ReplayAcc {
    uint64_t pushes = 0, cands = 0, busy_ns = 0, units = 0;
};

struct Slot {
    PinBuf<Cand> cand;  // written by the GPU straight into mapped host memory
    Cand* d_cand = nullptr;
    DevBuf<uint32_t> d_cnt;
    PinBuf<uint32_t> h_cnt;
    bool copies_ordered = false;  // coarse chunks: the record copies are on the copy stream (ev_done follows them)
    // coarse filter: the chunk's candidates compacted in (column, row) order - in HBM (d_so_*), and the host copy the
    // control thread orders on the copy stream once the counts are known (exactly `total` records per array);
    // h_meta: [0, P) candidates per column, [P, 2P) their offsets, [2P] total, [2P + 1] survivor keys emitted,
    // [2P + 2, 2P + 3] narrow scans: the chunk's MAC-passing rows (64 bits; instead of a copy of h_tested)
    double* so_score = nullptr;   // host copies: a piece of the session's pinned record ring (fetch_records)
    uint64_t* so_kmer = nullptr;
    uint32_t* so_row = nullptr;
    size_t ring_end = 0;          // ring offset behind this chunk's records (where the ring is free again once it is replayed)
    DevBuf<double> d_so_score;
    DevBuf<uint64_t> d_so_kmer;
    DevBuf<uint32_t> d_so_row;
    DevBuf<uint32_t> d_meta;
    PinBuf<uint32_t> h_meta;
    uint32_t* h_meta_dev = nullptr;  // the mapped buffers' device addresses (launch_chunk_tail)
    double* h_thr_dev = nullptr;
    unsigned long long* h_tested_dev = nullptr;
    PinBuf<double> h_thr;  // the device's thresholds behind this chunk (columns in select mode prune their pools with them, scan_lazy.cpp)
    bool tie_check = false;          // columns in select mode look at their pools for ties after this chunk (scan_lazy.cpp)
    bool tested_in_meta = false;     // the chunk's MAC-passing rows came in h_meta (narrow scans), not in h_tested
    hipEvent_t ev_counts = nullptr;  // compute stream: compaction done, h_meta copied
    DevBuf<unsigned long long> d_tested;
    PinBuf<unsigned long long> h_tested;
    hipEvent_t ev_sq0 = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr, ev_mid = nullptr;
    bool used_coarse = false;
    int coarse_mode = 0;
    double cand_est = 0;  // candidates the chunk was planned for (sum of topn x rows / rows before it)
    const uint64_t* rows = nullptr;
    uint64_t first_row = 0, n_rows = 0;
    bool squeezed = false, busy = false;
};

}  // namespace kgwas

using namespace kgwas;

struct kgwas_scan {
    int device = 0;
    uint64_t S_f = 0, S = 0, W_f = 0, W_m = 0, L = 0, n_pheno = 0, min_count = 0;
    std::vector<uint64_t> col, topn;
    std::vector<float> Y;
    bool direct = false;
    uint32_t kernel_used = 0;
    bool record_history = false;  // mode 1: every effective push is logged (hist)
    size_t history_ring = 0;      // mode 2: each heap keeps its last history_ring evictions instead (heap.h)
    uint64_t chunk_max = 0, dense_rows = 0, dense_chunk = 0;
    uint32_t cap = 0;
    uint64_t max_topn = 0;
    uint32_t nb_full = 0;  // leading 128-sample blocks the MFMA scorer may read unmasked

    hipStream_t stream = nullptr, copy_stream = nullptr;  // copy_stream: candidate records HBM -> host
    hipEvent_t ev_user = nullptr, ev_ds = nullptr, ev_d0 = nullptr, ev_d1 = nullptr;  // caller sync + dense-chunk timing
    hipEvent_t ev_dcopy = nullptr;     // the dense start's scores are in host memory (their copy runs on copy_stream, run_dense)
    bool dense_copy_pending = false;   // ... and nobody has waited for it yet (wait_dense_copy)
    DevBuf<uint32_t> d_dmask, d_colmap, d_sq;
    DevBuf<float> d_Yperm, d_Ymfma, d_sums;
    DevBuf<double> d_thr;
    PinBuf<double> h_thr;  // two halves, alternated, so an in-flight upload is never overwritten
    uint32_t thr_flip = 0;
    // device-side threshold tracking (thr_update_kernel)
    DevBuf<uint32_t> d_hist, d_hist_base;
    PinBuf<uint32_t> h_hist_base;
    DevBuf<uint64_t> d_topn;
    DevBuf<double> d_thr_host, d_thr_redo;
    PinBuf<double> h_thr_redo;
    bool hist_ready = false;
    uint64_t rows_submitted = 0;  // rows handed to the GPU (replayed or still in flight)
    // coarse int8 filter (sparse phase)
    bool coarse = false;
    uint32_t coarse_T = 0, n_kgroups = 0;  // coarse_T: most operand tiles the LDS can hold
    // Operand sets of the filter: mode[0] = one int8 slice per column (half the matrix work, ~2.5 survivors per
    // candidate), mode[1] = two slices (~1). Both may be resident; each chunk picks one (pick_coarse_mode).
    struct CoarsePart {  // one launch of the filter: n_lgroups LDS groups of T operand tiles over a range of columns
        uint32_t T = 0, n_lgroups = 0;
        uint32_t stream = 0;          // operand-streaming form (score_mxs.hip): 1 + launch_mxs's `form`; T = column tiles per column group
        uint32_t ng = 1;              // ... column groups per block (n_lgroups then counts operand groups of ng column groups)
        DevBuf<int8_t> d_Bq;
        DevBuf<CoarseCol> d_cols;
    };
    struct CoarseMode {
        bool ready = false;
        // block-scaled filter (score_mx.hip): FP6 (+ FP4 / FP6) slices instead of int8 ones; part[].T = column tiles
        bool mx = false;
        uint32_t mx_full = 0, mx_quarter = 0, mx_s1_fp6 = 0, mx_scale0 = 0;
        uint32_t slices = 0, n_parts = 0;
        uint32_t tile_slices = 0;  // operand tiles a row is multiplied with, all parts and groups
        double tile_slices_eq = 0;  // the same in int8 tile-slice equivalents (0: tile_slices as it is): what pick_coarse_mode compares
        float eg_max = 0, rall_max = 0, rmax_max = 0;  // row error term, maxima over the columns (kernels.h)
        // Full LDS groups first; columns that would only fill part of another full-size group go into a second launch
        // with as few tiles as they need (201 columns at 2048 samples: 3 groups x 4 tiles + 1 tile instead of 4 x 4).
        CoarsePart part[2];
    } cmode[2];
    // dense start overlapped with the first sparse chunks: thresholds selected on the device (launch_dense_select)
    DevBuf<double> d_sel;
    PinBuf<double> h_sel;
    DevBuf<uint32_t> d_sel_info;
    PinBuf<uint32_t> h_sel_info;
    bool sel_valid = false;  // h_sel holds the minima the heaps will have once the pending dense rows are pushed
    double infl_obs[2] = {4.0, 1.1};  // survivors per candidate of the last finished chunk of each mode
    double mode_k = 0.09;
    std::chrono::steady_clock::time_point t_feed0;  // KGWAS_TRACE: the timeline's origin (start of the current feed)
    double t_ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_feed0).count(); }
    uint64_t sum_topn = 0;  // over the columns
    // narrow filter (1-3 columns, score_narrow.hip): replaces coarse_kernel in the same pipeline
    bool narrow = false;
    bool narrow_pack1 = false;  // one column, replicated in the operand's four column slots (NarrowArgs::pack1)
    DevBuf<uint8_t> d_Bn;
    DevBuf<NarrowCol> d_ncols;
    DevBuf<unsigned long long> d_bitmap;  // survivors of the chunk being filtered: [n_pheno][bitmap_words]
    uint64_t bitmap_words = 0;
    DevBuf<uint32_t> d_bm_blocks;                  // block counts of launch_bitmap_keys / the narrow filter's per-segment counts
    DevBuf<unsigned long long> d_bm_mask;          // launch_bitmap_keys: per (column, block) the threads whose words hold a survivor
    bool bitmap_clean = false;                     // wide path: the key launches leave the bitmap all zero, so a chunk's prep launch has nothing to zero
    uint64_t slack_rows = 0;                       // rows of the feed's device buffer behind the chunk being submitted
    // survivors of the chunk being filtered: bitmap [column][64-row word], then row-ordered keys per column with each
    // column's range; shared by all chunks (consumed by the re-score kernel in stream order)
    DevBuf<uint32_t> d_surv_sorted, d_surv_cnt, d_surv_off, d_key_count;  // row-ordered keys per column, the columns' ranges, the total
    DevBuf<uint32_t> d_tile_pref, d_tile_cnt;  // tiles of 256 survivors (launch_rescore)
    DevBuf<double> d_tmp_score;                            // exact score of every survivor (-inf: not a candidate)
    uint32_t key_slots = 0;  // capacity of the key list = n_pheno * cap
    uint32_t row_key_bits = 32;
    // --pattern_counter
    bool count_patterns = false;
    DevBuf<uint64_t> d_pat;                // pattern hashes of the tested rows seen so far
    DevBuf<unsigned long long> d_pat_cnt;  // how many
    uint64_t pat_upper = 0;                // host-side upper bound of that count (rows fed)
    Slot slot[MAX_SLOTS];
    // Host copies of the coarse chunks' candidate records: ONE pinned ring, a chunk takes exactly its 20 B x candidates
    // when its counts are known and gives them back when it is replayed (FIFO). Slots used to own worst-case buffers
    // (cap x P records each), which capped the chunks in flight at 12 for 201 columns: with host and GPU level the GPU
    // then idled while the host digested the ramp.
    PinBuf<uint8_t> ring;
    uint8_t* ring_dev = nullptr;  // the ring's address on the device (mapped)
    bool streamed_feed = false;   // the rows of the current feed arrive over PCIe while it runs (ingest_run): fetch_records
    double* h_dense_dev = nullptr;  // device addresses of h_dense / h_n1 / h_kmer (mapped)
    uint32_t* h_n1_dev = nullptr;
    uint64_t* h_kmer_dev = nullptr;
    size_t ring_size = 0, ring_head = 0, ring_tail = 0;  // used: [tail, head) circularly; head == tail: empty
    // Columns in select mode keep their sparse chunks' records where the GPU wrote them (LazyCol::Seg): while this is set the
    // ring is filled linearly and nothing is given back - from one kgwas_scan_reset to the next, across feeds. Should it run full
    // (ring_to_recycling, scan_gpu.cpp), the logs take copies of what they refer to and the ring recycles as it does for
    // sessions that replay every column; from then on the workers copy each chunk's records into the logs.
    std::atomic<bool> ring_keep{false};
    size_t ring_feed_start = 0;  // ring_head when the current feed began
    uint64_t ring_freed = 0;                              // chunks (of this feed) whose records have been given back
    Slot redo;  // coarse mode: the only slot with exact-scorer candidate records (synchronous overflow re-runs)
    int n_slots = MAX_SLOTS;  // as many as fit 1 GiB of mapped pinned candidate memory (at least 4)
    // dense mode
    DevBuf<double> d_dense;
    DevBuf<uint32_t> d_n1;
    DevBuf<uint64_t> d_kmer;
    PinBuf<double> h_dense;
    PinBuf<uint32_t> h_n1;
    PinBuf<uint64_t> h_kmer;
    DevBuf<unsigned long long> d_tested_dense;
    // host / file ingest (kgwas_scan_feed_host, kgwas_scan_feed_table): three pinned pieces filled by a producer
    // thread, two device pieces, a copy stream; piece k+1 is read and copied while piece k is scored and replayed
    DevBuf<uint64_t> d_stage;  // kgwas_scan_scores_dense staging
    Ingest ingest;

    // all heaps' arrays in one 2 MB-aligned, MADV_HUGEPAGE arena (make_heaps); declared before the heaps: destroyed after
    struct HugeArena {
        void* p = nullptr;
        size_t bytes = 0;
        ~HugeArena() { free(p); }
    } heap_arena;
    std::unique_ptr<std::pmr::monotonic_buffer_resource> heap_mr;
    std::vector<BestHeap> heaps;
    // columns whose top-N is selected, not replayed (scan_lazy.cpp): lazy[j].on; lazy_any: some column may be
    bool lazy_enabled = false;
    // KGWAS_DEBUG_RESIDUALS (test hook): y_i - c - (what the slices of a form encode), per form: [0] one-slice set, [1] two-slice
    // set (block-scaled or int8, whichever was built), [2] narrow filter; [column * S + sample]
    bool dbg_keep_resid = false;
    std::vector<double> dbg_resid[3];
    // record_history = 2 sessions (the later shards of a cross-shard merge): their columns stay in select mode whatever their
    // ties - such a session is asked for its final minima and for its records above a threshold (kgwas_scan_lowest,
    // kgwas_scan_history_above: both served from the logs), not for result lists
    bool lazy_log_mode = false;
    std::vector<LazyCol> lazy;
    std::atomic<bool> lazy_any{false};
    std::atomic<uint64_t> n_selected{0}, n_unselected{0}, lazy_pushes{0};
    uint64_t tie_check_rows = 0;  // rows_submitted at which the next chunk is flagged for a tie check (geometric: x 1.25)
    std::vector<History> hist;
    std::vector<uint64_t> exp_kmer, exp_row;  // scratch of kgwas_scan_history_above / kgwas_scan_heaps_export
    std::vector<double> exp_score;
    std::vector<std::vector<uint64_t>> keys;  // per-column sort scratch for the replay
    bool trace = false;                       // KGWAS_TRACE=1: one stderr line per sparse chunk
    std::vector<double> col_ms;               // trace only: replay time per column of the last chunk
    bool all_full = false;
    uint64_t rows_done = 0;  // rows whose replay is complete
    std::unique_ptr<Pool> pool;
    // Streaming replay (feed_device_impl): columns in n_groups groups, (chunk, group) work units.
    struct alignas(64) GroupState {
        std::atomic<uint64_t> done{0};  // chunks of this feed the group has replayed = the next one it must take
        std::atomic<uint32_t> busy{0};  // a worker is on it
    };
    // The session's own groups are restored at every feed; during a feed a group that has fallen behind while other
    // workers have nothing left to do is SPLIT by the worker that holds it into single-column groups that anybody
    // takes (split_group, scan_replay.cpp), so n_groups grows and the arrays below have room for one group per column more.
    std::atomic<size_t> n_groups{1};
    size_t n_groups0 = 1;
    std::vector<std::vector<uint32_t>> grp_cols;   // columns of group g (at most BestHeap::MAX_LOCKSTEP)
    std::vector<std::vector<uint32_t>> grp_cols0;  // as the session was created
    std::vector<int> grp_home;                     // the worker that owns group g at the start of a feed, -1: floating (anybody takes it)
    std::unique_ptr<std::atomic<int>[]> grp_owner; // ... and now
    std::unique_ptr<GroupState[]> gstate;
    std::unique_ptr<std::atomic<uint32_t>[]> slot_left;  // [n_slots] COLUMNS that have not replayed the slot's chunk yet
    std::atomic<int> rp_hungry{0};                 // workers that found no unit to take the last time they looked
    std::atomic<bool> rp_all_published{false};     // the feed's last chunk is published: whoever is idle now stays idle
    std::atomic<uint64_t> rp_final_pub{0};         // the feed's chunk count, stored BEFORE rp_all_published (pop_ahead's "complete")
    // kgwas_scan_expect_finish: idle workers pop complete columns at the end of the (last) feed (scan_replay.cpp, pop_ahead)
    std::atomic<bool> final_feed{false};
    bool final_feed_next = false;                  // set by the hint, taken by the next feed
    std::unique_ptr<std::atomic<uint8_t>[]> col_popped;  // [n_pheno] 2: res_* of the column are those of its heap as it stands (1: being made)
    std::atomic<uint64_t> n_popped_ahead{0};
    std::mutex split_mu;
    bool split_lagging = true;                     // KGWAS_SPLIT_LAGGING=0: groups stay whole
    uint64_t float_lead = 2;                       // KGWAS_FLOAT_LEAD=n: a home group n chunks behind the foremost one floats (0: never)
    std::atomic<uint64_t> n_floated{0};
    int dbg_slow_worker = -1, dbg_slow_pct = 0, dbg_slow_min_us = 0;  // KGWAS_DEBUG_SLOW_WORKER=w:pct[:min_us] - worker w idles pct % of every unit's time on top (a busy co-tenant on its CPU), at least min_us microseconds (tests: a lag that does not depend on how long a unit takes)
    std::atomic<uint64_t> n_splits{0};
    std::atomic<uint64_t> seq_submitted{0}, seq_published{0}, seq_replayed{0};
    std::atomic<bool> rp_quit{false}, rp_failed{false};
    std::atomic<bool> rp_drain{false};  // with rp_quit: make the result lists of the columns that have none yet, then leave (scan_replay.cpp)
    std::atomic<int> rp_idle{0};
    std::mutex rp_mu;
    std::condition_variable rp_cv_work, rp_cv_done;
    std::function<void(size_t)> rp_fn;
    ReplayAcc rp_acc;  // sums over the workers of the current streaming replay
    std::atomic<uint64_t> prof_scan{0}, prof_heap{0};  // KGWAS_TRACE: TSC ticks in the record scans / in the heap updates
    uint64_t rp_max_busy_ns = 0, rp_min_busy_ns = ~0ull;
    kgwas_scan_stats st{};
    bool finished = false;
    std::vector<std::vector<uint64_t>> res_kmer, res_row;
    std::vector<std::vector<double>> res_score;

    ~kgwas_scan() {
        (void)hipSetDevice(device);
        auto drop_events = [](Slot& s) {
            if (s.ev_sq0) (void)hipEventDestroy(s.ev_sq0);
            if (s.ev_k0) (void)hipEventDestroy(s.ev_k0);
            if (s.ev_k1) (void)hipEventDestroy(s.ev_k1);
            if (s.ev_done) (void)hipEventDestroy(s.ev_done);
            if (s.ev_mid) (void)hipEventDestroy(s.ev_mid);
            if (s.ev_counts) (void)hipEventDestroy(s.ev_counts);
        };
        for (auto& s : slot) drop_events(s);
        drop_events(redo);
        if (ev_user) (void)hipEventDestroy(ev_user);
        if (ev_ds) (void)hipEventDestroy(ev_ds);
        if (ev_dcopy) (void)hipEventDestroy(ev_dcopy);
        if (ev_d0) (void)hipEventDestroy(ev_d0);
        if (ev_d1) (void)hipEventDestroy(ev_d1);
        if (stream) (void)hipStreamDestroy(stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

namespace kgwas {
// ---- scan_host.cpp
void check_device(int device);
void make_heaps(kgwas_scan* s);
// ---- scan_gpu.cpp: the GPU side of a chunk
void fill_args(kgwas_scan* s, ScoreArgs& a, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row, bool squeezed);
void hash_patterns(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows);
void upload_thresholds(kgwas_scan* s);
void refresh_full(kgwas_scan* s);
void run_dense(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row, double* scores_out, uint32_t* popcnt_out,
               bool replay, bool select = false);
void wait_dense_copy(kgwas_scan* s);
void dense_fill(kgwas_scan* s, uint64_t n_rows, uint64_t first_row, std::chrono::steady_clock::time_point td0,
                const std::function<void()>* meanwhile = nullptr);
int pick_coarse_mode(const kgwas_scan* s);
void submit_sparse(kgwas_scan* s, Slot& sl, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row, bool count_hist);
bool chunk_complete(kgwas_scan* s, Slot& sl);
uint64_t next_sparse_chunk(const kgwas_scan* s);
void wait_event(kgwas_scan* s, hipEvent_t ev);
bool fetch_records(kgwas_scan* s, Slot& sl, uint64_t seq);
void ring_to_recycling(kgwas_scan* s, uint64_t replayed);
// ---- scan_lazy.cpp: selection instead of replay
uint64_t lazy_materialize(kgwas_scan* s, size_t j);
void lazy_materialize_all(kgwas_scan* s);
void lazy_finish_column(kgwas_scan* s, size_t j, bool known_tie = false);
void lazy_reset(kgwas_scan* s);
bool lazy_lowest(kgwas_scan* s, size_t j, double* lowest, bool* full);
// ---- scan_replay.cpp: the host side
void replay_group(kgwas_scan* s, Slot& sl, size_t g, ReplayAcc& acc);
void add_replay_stats(kgwas_scan* s, const ReplayAcc& a);
void replay_worker(kgwas_scan* s, size_t w);
void feed_device_impl(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row);
}  // namespace kgwas
