// multiscan.cpp — the association scan and the kinship accumulation over several GPUs of one node, inside ONE
// process: the reference's caller runs a single associate_kmers / emma_kinship_kmers binary (kmers_gwas.py:133-148),
// so the row-sharding of SURVEY.md section 8e has to live behind that binary's boundary.
//
//   rows      contiguous shards in file order, shard g on devices[g] (global row order = concatenation of shards)
//   scan      one scan session + one host thread per shard, no data-path exchange; every session has its own replay
//             workers (the caller's thread budget divided by the shard count)
//   merge     shard 0's heaps after its rows ARE the global heaps after those rows. A later shard g contributes its
//             effective-push history filtered by  score > max(final minima of the full heaps of shards < g)  (anything
//             else add_association rejects whenever it arrives): kgwas_scan_history_above from each later session,
//             one kgwas_scan_absorb into shard 0's session, shards in row order (exactly the sequence of
//             add_association calls a single scan would make that can still change a heap)
//   kinship   integer partial sums (Hamming counts, rows used) of the shards add
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

using namespace kgwas;

namespace kgwas {
// scan_host.cpp: the pattern hashes a session has collected (device buffer, how many), for the cross-shard distinct count
void scan_patterns_peek(kgwas_scan* s, const uint64_t** d_hashes, uint64_t* n, int* device);
hipError_t count_distinct_u64(uint64_t* keys, uint64_t n, uint64_t* result, hipStream_t st);
unsigned usable_cpus_quota();  // scan_host.cpp: the cgroup CPU quota if there is one, else the hardware thread count
}  // namespace kgwas

struct kgwas_multiscan {
    std::vector<int32_t> dev;
    std::vector<kgwas_scan*> sess;
    // the caller's problem, kept for re-scans with the full push log (record_history = 1)
    kgwas_scan_params par{};
    std::vector<uint64_t> col, topn;
    std::vector<float> Y;
    uint32_t threads_per_shard = 1;
    double merge_ms = 0, scan_ms = 0;
    uint64_t rescans = 0, runs = 0;
    uint64_t rows_tested = 0, patterns = 0;
    // counters of the later shards' earlier runs (kgwas_scan_reset clears a session's statistics before every run)
    uint64_t prev_rows_fed = 0, prev_candidates = 0, prev_heap_pushes = 0, prev_chunks = 0, prev_score_launches = 0, prev_coarse_launches = 0;
    bool finished = false;
    ~kgwas_multiscan() {
        for (kgwas_scan* s : sess)
            if (s) kgwas_scan_destroy(s);
    }
};

namespace {

kgwas_scan* make_session(kgwas_multiscan* m, size_t g, uint32_t record_history) {
    kgwas_scan_params p = m->par;
    p.struct_size = sizeof(p);
    p.device = m->dev[g];
    p.col = m->col.data();
    p.Y = m->Y.data();
    p.topn = m->topn.data();
    p.host_threads = m->threads_per_shard;
    p.record_history = record_history;
    kgwas_scan* s = nullptr;
    if (kgwas_scan_create(&p, &s) != KGWAS_OK) throw Error(KGWAS_ERR_ARG, std::string("shard ") + std::to_string(g) + ": " + kgwas_last_error());
    return s;
}

// Run fn(g) for every shard on its own thread; the first failure is rethrown here with the thread's message.
template <class F>
void for_each_shard(size_t G, F&& fn) {
    std::vector<int> rc(G, KGWAS_OK);
    std::vector<std::string> msg(G);
    // (every shard needs a thread of its own - its session's calls block -: if the system has none to give, the threads that
    // exist are joined and the call fails)
    std::vector<std::thread> th;
    struct JoinAll {
        std::vector<std::thread>& t;
        ~JoinAll() {
            for (auto& x : t)
                if (x.joinable()) x.join();
        }
    } join_all{th};
    th.reserve(G);
    for (size_t g = 0; g < G; g++)
        th.emplace_back([&, g] {
            kgwas_name_this_thread("kgwas-shard");
            rc[g] = guarded([&] { fn(g); });
            if (rc[g] != KGWAS_OK) msg[g] = kgwas_last_error();
        });
    for (auto& t : th) t.join();
    for (size_t g = 0; g < G; g++)
        if (rc[g] != KGWAS_OK) throw Error(rc[g], "shard " + std::to_string(g) + ": " + msg[g]);
}

void ck(int rc) {
    if (rc != KGWAS_OK) throw Error(rc, kgwas_last_error());
}

// A later shard's session starts empty at every run call; its counters of the run before are kept in the totals.
void reset_shard(kgwas_multiscan* m, size_t g) {
    kgwas_scan_stats st;
    ck(kgwas_scan_get_stats(m->sess[g], &st));
    m->prev_rows_fed += st.rows_fed;
    m->prev_candidates += st.candidates;
    m->prev_heap_pushes += st.heap_pushes;
    m->prev_chunks += st.chunks;
    m->prev_score_launches += st.score_launches;
    m->prev_coarse_launches += st.coarse_launches;
    ck(kgwas_scan_reset(m->sess[g]));
}

// scan_shard(g, session): feed shard g's rows into the given session (used for the first scan and for re-scans).
template <class ScanShard>
void run_and_merge(kgwas_multiscan* m, ScanShard&& scan_shard) {
    const size_t G = m->sess.size();
    const uint64_t P = m->par.n_pheno;
    auto t0 = std::chrono::steady_clock::now();
    for_each_shard(G, [&](size_t g) { scan_shard(g, m->sess[g]); });
    auto t1 = std::chrono::steady_clock::now();
    m->scan_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (G > 1) {
        // thr[g][j] = max over shards h < g whose heap j is full of that heap's final minimum (-inf: no bound)
        const double ninf = -std::numeric_limits<double>::infinity();
        std::vector<double> run(P, ninf), low(P);
        std::vector<uint8_t> full(P);
        std::vector<uint64_t> counts((G - 1) * P);
        std::vector<const uint64_t*> kp(G - 1), rp(G - 1);
        std::vector<const double*> sp(G - 1);
        for (size_t g = 0; g < G; g++) {
            if (g > 0) {
                int rc = kgwas_scan_history_above(m->sess[g], run.data(), &counts[(g - 1) * P], &kp[g - 1], &sp[g - 1], &rp[g - 1]);
                if (rc == KGWAS_ERR_STATE) {
                    // the eviction ring was too short for this bound (shards of very different score levels): scan the
                    // shard again keeping the full push log
                    // (the replacement is created first: if that fails - out of memory for the log - the old session
                    // stays in place and the object remains usable)
                    kgwas_scan* fresh = make_session(m, g, 1);
                    kgwas_scan_destroy(m->sess[g]);
                    m->sess[g] = fresh;
                    scan_shard(g, m->sess[g]);
                    m->rescans++;
                    rc = kgwas_scan_history_above(m->sess[g], run.data(), &counts[(g - 1) * P], &kp[g - 1], &sp[g - 1], &rp[g - 1]);
                }
                ck(rc);
            }
            ck(kgwas_scan_lowest(m->sess[g], low.data(), full.data()));
            for (uint64_t j = 0; j < P; j++)
                if (full[j] && low[j] > run[j]) run[j] = low[j];  // (a NaN minimum gives no bound: the comparison fails)
        }
        ck(kgwas_scan_absorb(m->sess[0], G - 1, counts.data(), kp.data(), sp.data(), rp.data()));
    }
    m->merge_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    // per-call bookkeeping of the later shards (their sessions start empty again next time)
    for (size_t g = 1; g < G; g++) {
        kgwas_scan_stats st;
        ck(kgwas_scan_get_stats(m->sess[g], &st));
        m->rows_tested += st.rows_tested;
    }
}

}  // namespace

extern "C" {

int kgwas_multiscan_create(const kgwas_scan_params* p, const int32_t* devices, uint32_t n_devices, kgwas_multiscan** out) {
    return guarded([&] {
        if (!p || !out || !devices || n_devices == 0) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_create: null argument");
        if (p->struct_size != sizeof(kgwas_scan_params)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_params: size mismatch");
        if (!p->col || !p->Y || !p->topn || p->n_acc == 0 || p->n_pheno == 0) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_create: empty problem");
        std::unique_ptr<kgwas_multiscan> m(new kgwas_multiscan);
        m->dev.assign(devices, devices + n_devices);
        m->par = *p;
        m->col.assign(p->col, p->col + p->n_acc);
        m->topn.assign(p->topn, p->topn + p->n_pheno);
        m->Y.assign(p->Y, p->Y + p->n_pheno * p->n_acc);
        unsigned total = p->host_threads ? p->host_threads : usable_cpus_quota();
        m->threads_per_shard = std::max(1u, total / n_devices);
        m->sess.assign(n_devices, nullptr);
        // later shards keep each heap's last evictions (record_history = 2): all a merge needs from them
        for (size_t g = 0; g < n_devices; g++) m->sess[g] = make_session(m.get(), g, g == 0 ? p->record_history : 2u);
        *out = m.release();
    });
}

int kgwas_multiscan_run_table(kgwas_multiscan* m, kgwas_table* t, uint64_t row0, uint64_t n_rows) {
    return guarded([&] {
        if (!m || !t) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_run_table: null argument");
        const size_t G = m->sess.size();
        if (m->runs && m->par.count_patterns && G > 1)
            throw Error(KGWAS_ERR_STATE, "kgwas_multiscan: with count_patterns the rows must be handed over in one run call");
        m->runs++;
        for (size_t g = 1; g < G; g++) reset_shard(m, g);
        m->finished = false;
        run_and_merge(m, [&](size_t g, kgwas_scan* s) {
            const uint64_t lo = row0 + n_rows / G * g + std::min<uint64_t>(g, n_rows % G);
            const uint64_t cnt = n_rows / G + (g < n_rows % G ? 1 : 0);
            ck(kgwas_scan_feed_table(s, t, lo, cnt));
        });
    });
}

int kgwas_multiscan_run_device(kgwas_multiscan* m, const void* const* d_rows, const uint64_t* n_rows, const uint64_t* first_row) {
    return guarded([&] {
        if (!m || !d_rows || !n_rows || !first_row) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_run_device: null argument");
        const size_t G = m->sess.size();
        if (m->runs && m->par.count_patterns && G > 1)
            throw Error(KGWAS_ERR_STATE, "kgwas_multiscan: with count_patterns the rows must be handed over in one run call");
        m->runs++;
        for (size_t g = 1; g < G; g++) {
            if (first_row[g] != first_row[g - 1] + n_rows[g - 1]) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_run_device: shards must be contiguous, in row order");
            reset_shard(m, g);
        }
        m->finished = false;
        run_and_merge(m, [&](size_t g, kgwas_scan* s) { ck(kgwas_scan_feed_device(s, d_rows[g], n_rows[g], first_row[g], nullptr)); });
    });
}

int kgwas_multiscan_finish(kgwas_multiscan* m) {
    return guarded([&] {
        if (!m) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_finish: null");
        if (m->finished) return;
        ck(kgwas_scan_finish(m->sess[0]));
        if (m->par.count_patterns) {
            // distinct pattern hashes over ALL shards: gather every session's hashes on shard 0's device and count there
            const size_t G = m->sess.size();
            std::vector<const uint64_t*> src(G);
            std::vector<uint64_t> n(G);
            std::vector<int> dv(G);
            uint64_t total = 0;
            for (size_t g = 0; g < G; g++) {
                scan_patterns_peek(m->sess[g], &src[g], &n[g], &dv[g]);
                total += n[g];
            }
            m->patterns = 0;
            if (total) {
                KGWAS_HIP(hipSetDevice(dv[0]));
                uint64_t* all = nullptr;
                KGWAS_HIP(hipMalloc((void**)&all, total * sizeof(uint64_t)));
                uint64_t off = 0;
                hipError_t e = hipSuccess;
                for (size_t g = 0; g < G && e == hipSuccess; g++) {
                    if (n[g]) e = hipMemcpyPeer(all + off, dv[0], src[g], dv[g], n[g] * sizeof(uint64_t));
                    off += n[g];
                }
                uint64_t distinct = 0;
                if (e == hipSuccess) e = count_distinct_u64(all, total, &distinct, nullptr);
                (void)hipFree(all);
                KGWAS_HIP(e);
                m->patterns = distinct;
            }
        }
        m->finished = true;
    });
}

int kgwas_multiscan_result(kgwas_multiscan* m, uint64_t j, uint64_t* n, const uint64_t** kmer, const double** score, const uint64_t** row) {
    if (!m) {
        set_error("kgwas_multiscan_result: null");
        return KGWAS_ERR_ARG;
    }
    if (!m->finished) {
        set_error("call kgwas_multiscan_finish first");
        return KGWAS_ERR_STATE;
    }
    return kgwas_scan_result(m->sess[0], j, n, kmer, score, row);
}

int kgwas_multiscan_get_stats(const kgwas_multiscan* m, kgwas_scan_stats* total, kgwas_scan_stats* per_shard, double* scan_ms,
                              double* merge_ms, uint64_t* rescans) {
    return guarded([&] {
        if (!m || !total) throw Error(KGWAS_ERR_ARG, "kgwas_multiscan_get_stats: null");
        const size_t G = m->sess.size();
        kgwas_scan_stats t{};
        for (size_t g = 0; g < G; g++) {
            kgwas_scan_stats st;
            ck(kgwas_scan_get_stats(m->sess[g], &st));
            if (per_shard) per_shard[g] = st;
            if (g == 0) t = st;
            else {
                t.rows_fed += st.rows_fed;
                t.candidates += st.candidates;
                t.heap_pushes += st.heap_pushes;
                t.chunks += st.chunks;
                t.score_launches += st.score_launches;
                t.coarse_launches += st.coarse_launches;
                // times: the slowest shard's (the shards run side by side)
                t.score_kernel_ms = std::max(t.score_kernel_ms, st.score_kernel_ms);
                t.coarse_kernel_ms = std::max(t.coarse_kernel_ms, st.coarse_kernel_ms);
                t.replay_ms = std::max(t.replay_ms, st.replay_ms);
                t.replay_cpu_ms += st.replay_cpu_ms;
            }
        }
        t.rows_fed += m->prev_rows_fed;
        t.candidates += m->prev_candidates;
        t.heap_pushes += m->prev_heap_pushes;
        t.chunks += m->prev_chunks;
        t.score_launches += m->prev_score_launches;
        t.coarse_launches += m->prev_coarse_launches;
        t.rows_tested += m->rows_tested;  // later shards' counts, summed at every merge (their sessions are reset)
        if (m->par.count_patterns && m->finished) t.patterns = m->patterns;
        *total = t;
        if (scan_ms) *scan_ms = m->scan_ms;
        if (merge_ms) *merge_ms = m->merge_ms;
        if (rescans) *rescans = m->rescans;
    });
}

void kgwas_multiscan_destroy(kgwas_multiscan* m) { delete m; }

// Kinship over several devices: contiguous row shards of the table, one session + thread per shard, integer partials added.
int kgwas_kinship_table_multi(const int32_t* devices, uint32_t n_devices, kgwas_table* t, uint64_t min_count, uint64_t* hamming,
                              uint64_t* n_used) {
    return guarded([&] {
        if (!devices || n_devices == 0 || !t || !hamming || !n_used) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_table_multi: null argument");
        uint64_t S_f = 0, n_rows = 0, W_f = 0;
        uint32_t k = 0;
        ck(kgwas_table_info(t, &S_f, &n_rows, &W_f, &k));
        const size_t G = n_devices;
        std::vector<std::vector<uint64_t>> H(G);
        std::vector<uint64_t> used(G, 0);
        for_each_shard(G, [&](size_t g) {
            const uint64_t lo = n_rows / G * g + std::min<uint64_t>(g, n_rows % G);
            const uint64_t cnt = n_rows / G + (g < n_rows % G ? 1 : 0);
            kgwas_kinship* kin = nullptr;
            ck(kgwas_kinship_create(devices[g], S_f, min_count, &kin));
            int rc = kgwas_kinship_feed_table(kin, t, lo, cnt);
            if (rc == KGWAS_OK) {
                H[g].resize(S_f * S_f);
                rc = kgwas_kinship_partials(kin, H[g].data(), &used[g]);
            }
            std::string msg = rc == KGWAS_OK ? "" : kgwas_last_error();
            kgwas_kinship_destroy(kin);
            if (rc != KGWAS_OK) throw Error(rc, msg);
        });
        *n_used = 0;
        memset(hamming, 0, S_f * S_f * sizeof(uint64_t));
        for (size_t g = 0; g < G; g++) {
            *n_used += used[g];
            for (uint64_t i = 0; i < S_f * S_f; i++) hamming[i] += H[g][i];
        }
    });
}

}  // extern "C"
