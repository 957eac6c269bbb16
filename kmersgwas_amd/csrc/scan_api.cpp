// scan_api.cpp — the C ABI of the scan session apart from its creation (include/kgwas.h): feeds (device / host / .table),
// finish and results, histories and heap states for the cross-shard merges, statistics; BestAssociationsHeap and the
// shard merge through the C ABI.
#include "scan_internal.h"

// Layout of merge messages (include/kgwas.h: kgwas_scan_*_msgs).
namespace {
struct MsgPlan {
    std::vector<uint64_t> start;  // word offset of message m in the buffer (n_msgs + 1)
    uint64_t total() const { return start.back(); }
};
// kgwas_scan_expect_finish's hint belongs to the NEXT public feed call only, whatever becomes of that call (an argument
// error, an empty feed, an exception half-way): it is taken here, at the call's entry, and never outlives the call.
struct FinishHint {
    kgwas_scan* s;
    bool given;
    explicit FinishHint(kgwas_scan* s_) : s(s_), given(s_ && s_->final_feed_next) {
        if (s) s->final_feed_next = false;
    }
    ~FinishHint() {
        if (s) s->final_feed_next = false;
    }
};

// cnt(j) = entries of column j. Fills msg_words and returns the layout.
template <class Cnt>
MsgPlan plan_msgs(const kgwas_scan* s, uint64_t n_msgs, const uint64_t* col0, const uint64_t* ncols, uint64_t* msg_words, Cnt cnt, const char* who) {
    MsgPlan p;
    p.start.assign(n_msgs + 1, 0);
    for (uint64_t m = 0; m < n_msgs; m++) {
        if (ncols[m] && (col0[m] >= s->n_pheno || ncols[m] > s->n_pheno - col0[m])) throw Error(KGWAS_ERR_ARG, std::string(who) + ": columns out of range");
        uint64_t T = 0;
        for (uint64_t c = 0; c < ncols[m]; c++) T += cnt(col0[m] + c);
        msg_words[m] = 1 + ncols[m] + 3 * T;
        p.start[m + 1] = p.start[m] + msg_words[m];
    }
    return p;
}
}  // namespace

extern "C" {

uint32_t kgwas_host_cpu_quota(void) { return usable_cpus(); }
uint32_t kgwas_abi_version(void) { return KGWAS_ABI_VERSION; }

int kgwas_device_count(int* n_devices) {
    return guarded([&] {
        if (!n_devices) throw Error(KGWAS_ERR_ARG, "null argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *n_devices = (e == hipSuccess) ? n : 0;
    });
}

int kgwas_scan_feed_device(kgwas_scan* s, const void* d_rows, uint64_t n_rows, uint64_t first_row, void* hip_stream) {
    return guarded([&] {
        FinishHint hint(s);
        if (!s || (!d_rows && n_rows)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_device: null argument");
        if (s->finished) throw Error(KGWAS_ERR_STATE, "scan already finished");
        KGWAS_HIP(hipSetDevice(s->device));
        s->final_feed_next = hint.given && n_rows > 0;
        // order our stream after whatever the caller queued on theirs (e.g. the generator kernel)
        KGWAS_HIP(hipEventRecord(s->ev_user, (hipStream_t)hip_stream));
        KGWAS_HIP(hipStreamWaitEvent(s->stream, s->ev_user, 0));
        feed_device_impl(s, reinterpret_cast<const uint64_t*>(d_rows), n_rows, first_row);
    });
}

namespace {

// Chunked, double-buffered ingest (ingest.h): piece k+1 is produced and copied while piece k is scored and
// replayed; results do not depend on the piece size (rows are scored in order, thresholds only ever lag).
void ingest_run(kgwas_scan* s, uint64_t n_rows, uint64_t first_row, const Ingest::Fill& fill, bool expect_finish) {
    if (n_rows == 0) {
        feed_device_impl(s, nullptr, 0, first_row);
        return;
    }
    struct Streamed {  // (fetch_records picks the records' way to the host by it)
        kgwas_scan* s;
        explicit Streamed(kgwas_scan* s_) : s(s_) { s->streamed_feed = true; }
        ~Streamed() { s->streamed_feed = false; }
    } streamed(s);
    // it is the LAST piece's feed that is the last one
    s->ingest.run(1 + s->W_f, n_rows, s->chunk_max, s->stream, fill,
                  [&](const uint64_t* d_rows, uint64_t row_off, uint64_t cnt) {
                      s->final_feed_next = expect_finish && row_off + cnt == n_rows;
                      feed_device_impl(s, d_rows, cnt, first_row + row_off);  // returns with the stream idle
                  });
}

}  // namespace

int kgwas_scan_feed_host(kgwas_scan* s, const uint64_t* rows, uint64_t n_rows, uint64_t first_row) {
    return guarded([&] {
        FinishHint hint(s);
        if (!s || (!rows && n_rows)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_host: null argument");
        if (s->finished) throw Error(KGWAS_ERR_STATE, "scan already finished");
        KGWAS_HIP(hipSetDevice(s->device));
        const uint64_t stride = 1 + s->W_f;
        s->ingest.file_feed_ = false;
        ingest_run(s, n_rows, first_row, [&](uint64_t* dst, uint64_t row_off, uint64_t cnt) {
            memcpy(dst, rows + row_off * stride, cnt * stride * 8);
        }, hint.given);
    });
}

int kgwas_scan_feed_table(kgwas_scan* s, kgwas_table* t, uint64_t row0, uint64_t n_rows) {
    return guarded([&] {
        FinishHint hint(s);
        if (!s || !t) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_table: null argument");
        if (s->finished) throw Error(KGWAS_ERR_STATE, "scan already finished");
        uint64_t n_acc = 0, t_rows = 0, wpr = 0;
        uint32_t k = 0;
        if (kgwas_table_info(t, &n_acc, &t_rows, &wpr, &k) != KGWAS_OK) throw Error(KGWAS_ERR_ARG, kgwas_last_error());
        if (n_acc != s->S_f) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_table: table and scan disagree on the accession count");
        if (row0 > t_rows || n_rows > t_rows - row0) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_table: out of range");
        KGWAS_HIP(hipSetDevice(s->device));
        s->ingest.file_feed_ = true;
        ingest_run(s, n_rows, row0, [&](uint64_t* dst, uint64_t row_off, uint64_t cnt) {
            if (kgwas_table_read_rows(t, row0 + row_off, cnt, dst) != KGWAS_OK) throw Error(KGWAS_ERR_IO, kgwas_last_error());
        }, hint.given);
    });
}

int kgwas_scan_expect_finish(kgwas_scan* s) {
    return guarded([&] {
        if (!s) throw Error(KGWAS_ERR_ARG, "kgwas_scan_expect_finish: null");
        s->final_feed_next = true;
    });
}

int kgwas_scan_finish(kgwas_scan* s) {
    return guarded([&] {
        if (!s) throw Error(KGWAS_ERR_ARG, "kgwas_scan_finish: null");
        if (s->finished) return;
        KGWAS_HIP(hipSetDevice(s->device));
        KGWAS_HIP(hipStreamSynchronize(s->stream));
        s->res_kmer.resize(s->n_pheno);
        s->res_row.resize(s->n_pheno);
        s->res_score.resize(s->n_pheno);
        // The columns are popped in batches of up to eight in lockstep where their sizes agree. A batch is one item of the
        // pool, and there are about two per worker: a worker whose CPU is shared with somebody else keeps only the batch it
        // has started, the pool's other workers take the rest (with one item per worker - seven columns each at 101 columns
        // on 16 workers - finish took 3-5 ms instead of 1.8 on boxes with busy neighbours).
        // (Batches are made of the columns w, w + T, ... whose heaps worker w replayed last and are laid out so that the
        // pool hands worker w its own ones first: item r * T + w is worker w's r-th batch.)
        // columns in select mode first (scan_lazy.cpp): a selection each - or, where the scores do not decide the result, the
        // replay of the column's log through its heap and its pops
        if (s->lazy_any.load(std::memory_order_relaxed)) {
            std::vector<size_t> lz;
            for (size_t j = 0; j < s->n_pheno; j++)
                if (s->lazy[j].on && s->col_popped[j].load(std::memory_order_acquire) != 2) lz.push_back(j);
            // (A column whose tie turns up only here needs the replay of its whole log: ~4 ms at N = 10 001 and 10^8 rows, against
            // 0.1 ms per selection. Looking at every pool first - keys only - so that those replays start at once was measured:
            // 6.5 ms against 6.1, the selections in front of a replay are too short to matter. KGWAS_FINISH_TRACE=1 prints each
            // column's parts.)
            // KGWAS_FINISH_THREADS=n: a wider team for this step than the session's replay pool. A rank of a multi-GPU job scans with
            // its share of the node's CPUs (two of sixteen at eight ranks), but when rank 0 finishes the MERGED columns the other
            // ranks wait: the columns that need the exact replay (~10 ms each at the north-star shard) then go side by side on the
            // CPUs nobody is using instead of queueing on two threads (bench.py sets it for rank 0).
            const unsigned ft = (unsigned)std::max<long long>(0, opt_int("KGWAS_FINISH_THREADS", 0));
            if (ft > s->pool->size() && lz.size() > s->pool->size()) {
                std::atomic<size_t> next(0);
                kgwas_run_on_threads((unsigned)std::min<size_t>(ft, lz.size()), "kgwas-finish", [&] {
                    for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < lz.size();) {
                        lazy_finish_column(s, lz[i]);
                        s->col_popped[lz[i]].store(2, std::memory_order_release);
                    }
                });
            } else {
                s->pool->parallel_for(lz.size(), [&](size_t i) {
                    lazy_finish_column(s, lz[i]);
                    s->col_popped[lz[i]].store(2, std::memory_order_release);
                });
            }
        }
        s->st.heap_pushes += s->lazy_pushes.exchange(0);
        s->st.columns_selected = (uint32_t)s->n_selected.exchange(0);  // (as of this finish)
        s->st.columns_replayed_at_finish = (uint32_t)s->n_unselected.exchange(0);
        const size_t Tw = s->pool->size();
        const size_t per = std::min<size_t>(8, std::max<size_t>(1, (s->n_pheno + 2 * Tw - 1) / (2 * Tw)));
        std::vector<std::vector<std::vector<size_t>>> of_worker(Tw);  // [worker][batch][column]
        size_t rounds = 0;
        for (size_t w = 0; w < Tw; w++) {
            for (size_t j = w; j < s->n_pheno; j += Tw) {
                if (s->col_popped[j].load(std::memory_order_acquire) == 2) continue;  // popped at the end of the last feed (kgwas_scan_expect_finish)
                auto& bs = of_worker[w];
                if (bs.empty() || bs.back().size() == per || s->heaps[bs.back()[0]].size() != s->heaps[j].size()) bs.emplace_back();
                bs.back().push_back(j);
            }
            rounds = std::max(rounds, of_worker[w].size());
        }
        std::vector<const std::vector<size_t>*> batches;
        for (size_t r = 0; r < rounds; r++)
            for (size_t w = 0; w < Tw; w++) {
                static const std::vector<size_t> none;
                batches.push_back(r < of_worker[w].size() ? &of_worker[w][r] : &none);
            }
        s->pool->parallel_for(batches.size(), [&](size_t b) {
            const std::vector<size_t>& cols = *batches[b];
            const size_t K = cols.size();
            if (!K) return;
            const BestHeap* hp[8];
            std::vector<uint64_t>*km[8], *rw[8];
            std::vector<double>* sc[8];
            for (size_t k = 0; k < K; k++) {
                hp[k] = &s->heaps[cols[k]];
                km[k] = &s->res_kmer[cols[k]];
                sc[k] = &s->res_score[cols[k]];
                rw[k] = &s->res_row[cols[k]];
            }
            BestHeap::pop_all_n((int)K, hp, km, sc, rw);
        });
        if (s->count_patterns) {
            unsigned long long n_hashes = 0;
            KGWAS_HIP(hipMemcpy(&n_hashes, s->d_pat_cnt.p, 8, hipMemcpyDeviceToHost));
            uint64_t distinct = 0;
            KGWAS_HIP(count_distinct_u64(s->d_pat.p, n_hashes, &distinct, s->stream));
            s->st.patterns = distinct;
        }
        s->finished = true;
    });
}

int kgwas_scan_result(kgwas_scan* s, uint64_t j, uint64_t* n, const uint64_t** kmer, const double** score,
                      const uint64_t** row) {
    return guarded([&] {
        if (!s || j >= s->n_pheno) throw Error(KGWAS_ERR_ARG, "kgwas_scan_result: bad argument");
        if (!s->finished) throw Error(KGWAS_ERR_STATE, "call kgwas_scan_finish first");
        if (n) *n = s->res_kmer[j].size();
        if (kmer) *kmer = s->res_kmer[j].data();
        if (score) *score = s->res_score[j].data();
        if (row) *row = s->res_row[j].data();
    });
}

int kgwas_scan_history(kgwas_scan* s, uint64_t j, uint64_t* n, const uint64_t** kmer, const double** score,
                       const uint64_t** row) {
    return guarded([&] {
        if (!s || j >= s->n_pheno) throw Error(KGWAS_ERR_ARG, "kgwas_scan_history: bad argument");
        if (!s->record_history) throw Error(KGWAS_ERR_STATE, "scan was created without record_history");
        History& h = s->hist[j];
        _mm_sfence();
        h.v_kmer.resize(h.n);
        h.v_score.resize(h.n);
        h.v_row.resize(h.n);
        for (size_t i = 0; i < h.n; i++) {
            h.v_kmer[i] = h.p[i].kmer;
            h.v_score[i] = h.p[i].score;
            h.v_row[i] = h.p[i].row;
        }
        if (n) *n = h.n;
        if (kmer) *kmer = h.v_kmer.data();
        if (score) *score = h.v_score.data();
        if (row) *row = h.v_row.data();
    });
}

int kgwas_scan_history_above(kgwas_scan* s, const double* thr, uint64_t* counts, const uint64_t** kmer,
                             const double** score, const uint64_t** row) {
    return guarded([&] {
        if (!s || !thr || !counts) throw Error(KGWAS_ERR_ARG, "kgwas_scan_history_above: null argument");
        if (!s->record_history && !s->history_ring) throw Error(KGWAS_ERR_STATE, "scan was created without record_history");
        const uint64_t P = s->n_pheno;
        if (s->lazy_log_mode && s->lazy_any.load(std::memory_order_relaxed)) {
            // Columns in select mode (scan_lazy.cpp): every record the column was shipped above thr, in row order - a superset of
            // its effective pushes above thr, and the same thing to the heap they are merged into: a record this shard's heap
            // rejected (score <= its minimum at that row) is rejected there too, where the minimum is at least as high. A column
            // that has been given its heap meanwhile (kgwas_scan_finish on a column with ties) answers from the heap's ring.
            const double ninf = -std::numeric_limits<double>::infinity();
            auto keepl = [&](uint64_t j, double sc) { return sc != ninf && (thr[j] == ninf || sc > thr[j]); };  // (-inf: a narrow chunk's placeholder)
            _mm_sfence();
            std::vector<std::vector<BestHeap::Rec>> recs(P);
            std::vector<char> ok(P, 1);
            s->pool->parallel_for(P, [&](size_t j) {
                if (!s->lazy[j].on) {
                    ok[j] = s->heaps[j].pushes_above(thr[j], recs[j]) ? 1 : 0;
                    counts[j] = recs[j].size();
                    return;
                }
                const LazyCol& h = s->lazy[j];
                uint64_t c = 0;
                h.for_each([&](double sc, uint64_t, uint64_t) { c += keepl(j, sc) ? 1 : 0; });
                counts[j] = c;
            });
            for (uint64_t j = 0; j < P; j++)
                if (!ok[j]) throw Error(KGWAS_ERR_STATE, "record_history = 2: column " + std::to_string(j) + " needs evictions that left its ring (raise KGWAS_HISTORY_RING, or use record_history = 1)");
            std::vector<uint64_t> off(P + 1, 0);
            for (uint64_t j = 0; j < P; j++) off[j + 1] = off[j] + counts[j];
            s->exp_kmer.resize(off[P]);
            s->exp_score.resize(off[P]);
            s->exp_row.resize(off[P]);
            s->pool->parallel_for(P, [&](size_t j) {
                uint64_t o = off[j];
                if (!s->lazy[j].on) {
                    for (const BestHeap::Rec& r : recs[j]) {
                        s->exp_kmer[o] = r.kmer;
                        s->exp_score[o] = r.score;
                        s->exp_row[o] = r.row;
                        o++;
                    }
                    return;
                }
                const LazyCol& h = s->lazy[j];
                h.for_each([&](double sc, uint64_t km, uint64_t rw) {
                    if (!keepl(j, sc)) return;
                    s->exp_kmer[o] = km;
                    s->exp_score[o] = sc;
                    s->exp_row[o] = rw;
                    o++;
                });
            });
            if (kmer) *kmer = s->exp_kmer.data();
            if (score) *score = s->exp_score.data();
            if (row) *row = s->exp_row.data();
            return;
        }
        lazy_materialize_all(s);
        if (s->history_ring) {  // mode 2: the heap's entries and its last evictions above thr, put in row order
            std::vector<std::vector<BestHeap::Rec>> recs(P);
            std::vector<char> ok(P, 1);
            s->pool->parallel_for(P, [&](size_t j) { ok[j] = s->heaps[j].pushes_above(thr[j], recs[j]) ? 1 : 0; });
            for (uint64_t j = 0; j < P; j++)
                if (!ok[j])
                    throw Error(KGWAS_ERR_STATE, "record_history = 2: column " + std::to_string(j) + " needs evictions that left its ring of " +
                                                     std::to_string(ring_size(s->history_ring, s->topn[j])) + " (raise KGWAS_HISTORY_RING, or use record_history = 1)");
            std::vector<uint64_t> off(P + 1, 0);
            for (uint64_t j = 0; j < P; j++) {
                counts[j] = recs[j].size();
                off[j + 1] = off[j] + counts[j];
            }
            s->exp_kmer.resize(off[P]);
            s->exp_score.resize(off[P]);
            s->exp_row.resize(off[P]);
            s->pool->parallel_for(P, [&](size_t j) {
                uint64_t o = off[j];
                for (const BestHeap::Rec& r : recs[j]) {
                    s->exp_kmer[o] = r.kmer;
                    s->exp_score[o] = r.score;
                    s->exp_row[o] = r.row;
                    o++;
                }
            });
            if (kmer) *kmer = s->exp_kmer.data();
            if (score) *score = s->exp_score.data();
            if (row) *row = s->exp_row.data();
            return;
        }
        // entries add_association could still accept after heaps whose minimum is thr[j]: score > thr[j]
        // (NaN scores never pass; thr = -inf keeps everything, NaN included, as the heap may not be full)
        auto keep = [&](uint64_t j, double sc) { return thr[j] == -std::numeric_limits<double>::infinity() || sc > thr[j]; };
        _mm_sfence();
        s->pool->parallel_for(P, [&](size_t j) {
            const History& h = s->hist[j];
            uint64_t c = 0;
            for (size_t i = 0; i < h.n; i++) c += keep(j, h.p[i].score) ? 1 : 0;
            counts[j] = c;
        });
        std::vector<uint64_t> off(P + 1, 0);
        for (uint64_t j = 0; j < P; j++) off[j + 1] = off[j] + counts[j];
        s->exp_kmer.resize(off[P]);
        s->exp_score.resize(off[P]);
        s->exp_row.resize(off[P]);
        s->pool->parallel_for(P, [&](size_t j) {
            const History& h = s->hist[j];
            uint64_t o = off[j];
            for (size_t i = 0; i < h.n; i++)
                if (keep(j, h.p[i].score)) {
                    s->exp_kmer[o] = h.p[i].kmer;
                    s->exp_score[o] = h.p[i].score;
                    s->exp_row[o] = h.p[i].row;
                    o++;
                }
        });
        if (kmer) *kmer = s->exp_kmer.data();
        if (score) *score = s->exp_score.data();
        if (row) *row = s->exp_row.data();
    });
}

int kgwas_scan_heaps_export(kgwas_scan* s, uint64_t n_cols, const uint64_t* cols, uint64_t* sizes, const uint64_t** kmer,
                            const double** score, const uint64_t** row) {
    return guarded([&] {
        if (!s || (n_cols && (!cols || !sizes))) throw Error(KGWAS_ERR_ARG, "kgwas_scan_heaps_export: null argument");
        lazy_materialize_all(s);  // (columns in select mode get their heaps first: scan_lazy.cpp)
        std::vector<uint64_t> off(n_cols + 1, 0);
        for (uint64_t c = 0; c < n_cols; c++) {
            if (cols[c] >= s->n_pheno) throw Error(KGWAS_ERR_ARG, "kgwas_scan_heaps_export: column out of range");
            sizes[c] = s->heaps[cols[c]].size();
            off[c + 1] = off[c] + sizes[c];
        }
        s->exp_kmer.resize(off[n_cols]);
        s->exp_score.resize(off[n_cols]);
        s->exp_row.resize(off[n_cols]);
        s->pool->parallel_for(n_cols, [&](size_t c) {
            s->heaps[cols[c]].export_state(s->exp_kmer.data() + off[c], s->exp_score.data() + off[c], s->exp_row.data() + off[c]);
        });
        if (kmer) *kmer = s->exp_kmer.data();
        if (score) *score = s->exp_score.data();
        if (row) *row = s->exp_row.data();
    });
}

int kgwas_scan_heaps_import(kgwas_scan* s, uint64_t n_cols, const uint64_t* cols, const uint64_t* sizes, const uint64_t* kmer,
                            const double* score, const uint64_t* row) {
    return guarded([&] {
        if (!s || (n_cols && (!cols || !sizes))) throw Error(KGWAS_ERR_ARG, "kgwas_scan_heaps_import: null argument");
        lazy_materialize_all(s);  // (columns in select mode get their heaps first: scan_lazy.cpp)
        for (uint64_t c = 0; c < n_cols; c++) {
            if (cols[c] >= s->n_pheno) throw Error(KGWAS_ERR_ARG, "kgwas_scan_heaps_import: column out of range");
            if (sizes[c] && (!kmer || !score || !row)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_heaps_import: null data");
        }
        std::vector<uint64_t> off(n_cols + 1, 0);
        for (uint64_t c = 0; c < n_cols; c++) off[c + 1] = off[c] + sizes[c];
        s->pool->parallel_for(n_cols, [&](size_t c) {
            s->heaps[cols[c]].import_state((size_t)sizes[c], kmer + off[c], score + off[c], row + off[c]);
        });
        s->finished = false;
        for (uint64_t j = 0; j < s->n_pheno; j++) s->col_popped[j].store(0);
        refresh_full(s);
        // The device's thresholds and score histograms describe the rows behind the heaps that were just replaced:
        // start them again from the imported minima before anything else is fed (the next sparse chunk re-bases the
        // histograms, start_histograms).
        KGWAS_HIP(hipSetDevice(s->device));
        s->hist_ready = false;
        upload_thresholds(s);
    });
}

// ---- merge messages (include/kgwas.h): [words that follow][counts: n][kmer: T][score bits: T][row: T] --------------
int kgwas_scan_history_above_msgs(kgwas_scan* s, const double* thr, uint64_t n_msgs, const uint64_t* msg_col0,
                                  const uint64_t* msg_ncols, uint64_t* out, uint64_t cap_words, uint64_t* msg_words) {
    return guarded([&] {
        if (!s || !thr || (n_msgs && (!msg_col0 || !msg_ncols || !msg_words))) throw Error(KGWAS_ERR_ARG, "kgwas_scan_history_above_msgs: null argument");
        if (!s->record_history && !s->history_ring) throw Error(KGWAS_ERR_STATE, "scan was created without record_history");
        const bool from_logs = s->lazy_log_mode && s->lazy_any.load(std::memory_order_relaxed);  // (see kgwas_scan_history_above)
        if (!from_logs) lazy_materialize_all(s);
        const uint64_t P = s->n_pheno;
        std::vector<char> wanted(P, 0);
        for (uint64_t m = 0; m < n_msgs; m++) {  // (ranges first: nothing below loops over an unchecked count)
            if (msg_ncols[m] && (msg_col0[m] >= P || msg_ncols[m] > P - msg_col0[m])) throw Error(KGWAS_ERR_ARG, "kgwas_scan_history_above_msgs: column range out of bounds");
            for (uint64_t c = 0; c < msg_ncols[m]; c++) wanted[msg_col0[m] + c] = 1;
        }
        const double ninf = -std::numeric_limits<double>::infinity();
        auto keep = [&](uint64_t j, double sc) { return thr[j] == ninf || sc > thr[j]; };
        std::vector<std::vector<BestHeap::Rec>> recs(s->history_ring ? P : 0);
        std::vector<uint64_t> counts(P, 0);
        std::vector<char> ok(P, 1);
        if (!s->history_ring || from_logs) _mm_sfence();
        s->pool->parallel_for(P, [&](size_t j) {
            if (!wanted[j]) return;
            if (from_logs && s->lazy[j].on) {
                const LazyCol& h = s->lazy[j];
                uint64_t c = 0;
                h.for_each([&](double sc, uint64_t, uint64_t) { c += (sc != ninf && keep(j, sc)) ? 1 : 0; });
                counts[j] = c;
            } else if (s->history_ring) {  // mode 2: the heap's entries and its last evictions above thr, put in row order
                ok[j] = s->heaps[j].pushes_above(thr[j], recs[j]) ? 1 : 0;
                counts[j] = recs[j].size();
            } else {
                const History& h = s->hist[j];
                uint64_t c = 0;
                for (size_t i = 0; i < h.n; i++) c += keep(j, h.p[i].score) ? 1 : 0;
                counts[j] = c;
            }
        });
        for (uint64_t j = 0; j < P; j++)
            if (!ok[j])
                throw Error(KGWAS_ERR_STATE, "record_history = 2: column " + std::to_string(j) + " needs evictions that left its ring of " +
                                                 std::to_string(ring_size(s->history_ring, s->topn[j])) + " (raise KGWAS_HISTORY_RING, or use record_history = 1)");
        const MsgPlan plan = plan_msgs(s, n_msgs, msg_col0, msg_ncols, msg_words, [&](uint64_t j) { return counts[j]; }, "kgwas_scan_history_above_msgs");
        if (!out || cap_words < plan.total()) return;
        struct Job {
            uint64_t j, ok, os, orw;  // column, word offsets of its kmer / score / row runs
        };
        std::vector<Job> jobs;
        for (uint64_t m = 0; m < n_msgs; m++) {
            uint64_t* msg = out + plan.start[m];
            msg[0] = msg_words[m] - 1;
            uint64_t T = 0;
            for (uint64_t c = 0; c < msg_ncols[m]; c++) T += (msg[1 + c] = counts[msg_col0[m] + c]);
            uint64_t o = plan.start[m] + 1 + msg_ncols[m];
            for (uint64_t c = 0; c < msg_ncols[m]; c++) {
                const uint64_t j = msg_col0[m] + c;
                jobs.push_back(Job{j, o, o + T, o + 2 * T});
                o += counts[j];
            }
        }
        s->pool->parallel_for(jobs.size(), [&](size_t i) {
            const Job& b = jobs[i];
            uint64_t* k = out + b.ok;
            uint64_t* sc = out + b.os;
            uint64_t* rw = out + b.orw;
            uint64_t o = 0;
            auto put = [&](uint64_t kmer, double score, uint64_t row) {
                k[o] = kmer;
                std::memcpy(&sc[o], &score, 8);
                rw[o] = row;
                o++;
            };
            if (from_logs && s->lazy[b.j].on) {
                const LazyCol& h = s->lazy[b.j];
                h.for_each([&](double sc, uint64_t km, uint64_t rw) {
                    if (sc != ninf && keep(b.j, sc)) put(km, sc, rw);
                });
            } else if (s->history_ring) {
                for (const BestHeap::Rec& r : recs[b.j]) put(r.kmer, r.score, r.row);
            } else {
                const History& h = s->hist[b.j];
                for (size_t e = 0; e < h.n; e++)
                    if (keep(b.j, h.p[e].score)) put(h.p[e].kmer, h.p[e].score, h.p[e].row);
            }
        });
    });
}

int kgwas_scan_heaps_export_msgs(kgwas_scan* s, uint64_t n_msgs, const uint64_t* msg_col0, const uint64_t* msg_ncols,
                                 uint64_t* out, uint64_t cap_words, uint64_t* msg_words) {
    return guarded([&] {
        if (!s || (n_msgs && (!msg_col0 || !msg_ncols || !msg_words))) throw Error(KGWAS_ERR_ARG, "kgwas_scan_heaps_export_msgs: null argument");
        lazy_materialize_all(s);
        const MsgPlan plan = plan_msgs(s, n_msgs, msg_col0, msg_ncols, msg_words, [&](uint64_t j) { return (uint64_t)s->heaps[j].size(); }, "kgwas_scan_heaps_export_msgs");
        if (!out || cap_words < plan.total()) return;
        struct Job {
            uint64_t j, ok, os, orw;
        };
        std::vector<Job> jobs;
        for (uint64_t m = 0; m < n_msgs; m++) {
            uint64_t* msg = out + plan.start[m];
            msg[0] = msg_words[m] - 1;
            uint64_t T = 0;
            for (uint64_t c = 0; c < msg_ncols[m]; c++) T += (msg[1 + c] = s->heaps[msg_col0[m] + c].size());
            uint64_t o = plan.start[m] + 1 + msg_ncols[m];
            for (uint64_t c = 0; c < msg_ncols[m]; c++) {
                const uint64_t j = msg_col0[m] + c;
                jobs.push_back(Job{j, o, o + T, o + 2 * T});
                o += s->heaps[j].size();
            }
        }
        static_assert(sizeof(double) == sizeof(uint64_t), "score bit patterns travel as 64-bit words");
        s->pool->parallel_for(jobs.size(), [&](size_t i) {
            const Job& b = jobs[i];
            s->heaps[b.j].export_state(out + b.ok, reinterpret_cast<double*>(out + b.os), out + b.orw);
        });
    });
}

int kgwas_scan_debug_residuals(const kgwas_scan* s, uint32_t form, uint64_t column, double* out) {
    return guarded([&] {
        if (!s || !out || form > 2 || column >= s->n_pheno) throw Error(KGWAS_ERR_ARG, "kgwas_scan_debug_residuals: bad argument");
        if (!s->dbg_keep_resid) throw Error(KGWAS_ERR_STATE, "kgwas_scan_debug_residuals: the session was not created under KGWAS_DEBUG_RESIDUALS=1");
        memcpy(out, s->dbg_resid[form].data() + column * s->S, s->S * sizeof(double));
    });
}

int kgwas_scan_select_mode(const kgwas_scan* s, int* on) {
    return guarded([&] {
        if (!s || !on) throw Error(KGWAS_ERR_ARG, "kgwas_scan_select_mode: null argument");
        *on = s->lazy_enabled ? 1 : 0;
    });
}

int kgwas_scan_lowest(kgwas_scan* s, double* lowest, uint8_t* full) {
    return guarded([&] {
        if (!s || !lowest || !full) throw Error(KGWAS_ERR_ARG, "kgwas_scan_lowest: null argument");
        for (uint64_t j = 0; j < s->n_pheno; j++) {
            if (s->lazy[j].on) {  // select mode: the N-th largest score logged IS the heap's minimum (scan_lazy.cpp)
                bool f = false;
                if (lazy_lowest(s, j, &lowest[j], &f)) {
                    full[j] = f ? 1 : 0;
                    continue;
                }
                s->st.heap_pushes += lazy_materialize(s, j);
            }
            lowest[j] = s->heaps[j].lowest();
            full[j] = s->heaps[j].full() ? 1 : 0;
        }
    });
}

int kgwas_scan_absorb(kgwas_scan* s, uint64_t n_shards, const uint64_t* counts, const uint64_t* const* kmer,
                      const double* const* score, const uint64_t* const* row) {
    return guarded([&] {
        if (!s || (n_shards && (!counts || !kmer || !score || !row))) throw Error(KGWAS_ERR_ARG, "kgwas_scan_absorb: null argument");
        const uint64_t P = s->n_pheno;
        std::vector<std::vector<uint64_t>> off(n_shards, std::vector<uint64_t>(P + 1, 0));
        for (uint64_t g = 0; g < n_shards; g++)
            for (uint64_t j = 0; j < P; j++) off[g][j + 1] = off[g][j] + counts[g * P + j];
        std::atomic<uint64_t> pushes(0);
        s->pool->parallel_for(P, [&](size_t j) {
            BestHeap& h = s->heaps[j];
            uint64_t local = 0;
            if (s->lazy[j].on) {  // select mode: the later shards' records join the column's log (rows behind everything it holds)
                for (uint64_t g = 0; g < n_shards; g++) {
                    const uint64_t o = off[g][j], n = counts[g * P + j];
                    for (uint64_t i = 0; i < n; i++) s->lazy[j].add(kmer[g][o + i], score[g][o + i], row[g][o + i]);
                }
                return;
            }
            for (uint64_t g = 0; g < n_shards; g++) {  // shards in row order
                const uint64_t o = off[g][j], n = counts[g * P + j];
                for (uint64_t i = 0; i < n; i++)
                    if (h.add(kmer[g][o + i], score[g][o + i], (size_t)row[g][o + i])) {
                        local++;
                        if (s->record_history) s->hist[j].push(kmer[g][o + i], score[g][o + i], row[g][o + i]);
                    }
            }
            if (s->record_history) _mm_sfence();  // streaming stores of the history log
        pushes += local;
        });
        s->st.heap_pushes += pushes.load();
        s->finished = false;
        for (uint64_t j = 0; j < s->n_pheno; j++) s->col_popped[j].store(0);
        refresh_full(s);
        // feeding may go on after an absorb: the device thresholds follow the heaps (see kgwas_scan_heaps_import)
        KGWAS_HIP(hipSetDevice(s->device));
        s->hist_ready = false;
        upload_thresholds(s);
    });
}

int kgwas_scan_reset(kgwas_scan* s) {
    return guarded([&] {
        if (!s) throw Error(KGWAS_ERR_ARG, "kgwas_scan_reset: null");
        KGWAS_HIP(hipSetDevice(s->device));
        KGWAS_HIP(hipStreamSynchronize(s->stream));
        make_heaps(s);
        lazy_reset(s);
        for (uint64_t j = 0; j < s->n_pheno; j++) s->col_popped[j].store(0);
        s->final_feed_next = false;
        for (auto& h : s->hist) h.clear();
        s->all_full = false;
        s->hist_ready = false;
        s->rows_submitted = 0;
        s->pat_upper = 0;
        KGWAS_HIP(hipMemset(s->d_pat_cnt.p, 0, 8));
        s->rows_done = 0;
        s->finished = false;
        const kgwas_scan_stats old = s->st;
        s->st = kgwas_scan_stats{};
        s->st.kernel_used = old.kernel_used;
        s->st.direct_mode = old.direct_mode;
        s->st.coarse_mx = old.coarse_mx;
        s->st.coarse_mx_s1_fp6 = old.coarse_mx_s1_fp6;
        s->st.coarse_mx_steps = old.coarse_mx_steps;
        s->st.coarse_mx_stream = old.coarse_mx_stream;
        s->st.replay_threads = old.replay_threads;
        for (int mi = 0; mi < 2; mi++) {
            s->st.coarse_mode_tiles[mi] = old.coarse_mode_tiles[mi];
            s->st.coarse_mode_lgroups[mi] = old.coarse_mode_lgroups[mi];
            s->st.coarse_mode_tile_slices[mi] = old.coarse_mode_tile_slices[mi];
        }
    });
}

int kgwas_scan_get_stats(const kgwas_scan* s, kgwas_scan_stats* st) {
    return guarded([&] {
        if (!s || !st) throw Error(KGWAS_ERR_ARG, "kgwas_scan_get_stats: null");
        *st = s->st;
    });
}

void kgwas_scan_destroy(kgwas_scan* s) { delete s; }

int kgwas_scan_scores_dense(kgwas_scan* s, const void* rows, int rows_on_device, uint64_t n_rows, double* scores,
                            uint32_t* popcnt) {
    return guarded([&] {
        if (!s || (!rows && n_rows) || !scores || !popcnt) throw Error(KGWAS_ERR_ARG, "kgwas_scan_scores_dense: null argument");
        KGWAS_HIP(hipSetDevice(s->device));
        const uint64_t stride = 1 + s->W_f;
        const uint64_t piece = s->dense_rows;
        if (!rows_on_device && s->d_stage.n < piece * stride) s->d_stage.alloc(piece * stride);
        std::vector<double> tmp(s->n_pheno * piece);
        for (uint64_t pos = 0; pos < n_rows; pos += piece) {
            const uint64_t c = std::min<uint64_t>(piece, n_rows - pos);
            const uint64_t* d_rows;
            if (rows_on_device) {
                d_rows = reinterpret_cast<const uint64_t*>(rows) + pos * stride;
            } else {
                KGWAS_HIP(hipMemcpy(s->d_stage.p, reinterpret_cast<const uint64_t*>(rows) + pos * stride, c * stride * 8,
                                    hipMemcpyHostToDevice));
                d_rows = s->d_stage.p;
            }
            run_dense(s, d_rows, c, pos, tmp.data(), popcnt + pos, false);
            for (uint64_t j = 0; j < s->n_pheno; j++)
                memcpy(scores + j * n_rows + pos, tmp.data() + j * c, c * sizeof(double));
        }
    });
}

// ---- BestAssociationsHeap through the C ABI -------------------------------------------------
struct kgwas_heap {
    BestHeap h;
    explicit kgwas_heap(size_t n) : h(n) {}
};

int kgwas_heap_new(uint64_t max_results, kgwas_heap** out) {
    return guarded([&] {
        if (!out || max_results == 0) throw Error(KGWAS_ERR_ARG, "kgwas_heap_new: bad argument");
        require_heap_emulation();
        *out = new kgwas_heap((size_t)max_results);
    });
}
int kgwas_heap_add_many(kgwas_heap* h, const uint64_t* kmer, const double* score, const uint64_t* row, uint64_t n) {
    return guarded([&] {
        if (!h || (n && (!kmer || !score || !row))) throw Error(KGWAS_ERR_ARG, "kgwas_heap_add_many: null argument");
        for (uint64_t i = 0; i < n; i++) h->h.add(kmer[i], score[i], (size_t)row[i]);
    });
}
int kgwas_heap_size(const kgwas_heap* h, uint64_t* size, uint64_t* insertions, double* lowest) {
    return guarded([&] {
        if (!h) throw Error(KGWAS_ERR_ARG, "kgwas_heap_size: null");
        if (size) *size = h->h.size();
        if (insertions) *insertions = h->h.inserted();
        if (lowest) *lowest = h->h.lowest();
    });
}
int kgwas_heap_pop_all(const kgwas_heap* h, uint64_t* kmer, double* score, uint64_t* row) {
    return guarded([&] {
        if (!h) throw Error(KGWAS_ERR_ARG, "kgwas_heap_pop_all: null");
        std::vector<uint64_t> k, r;
        std::vector<double> sc;
        h->h.pop_all(k, sc, r);
        if (kmer) memcpy(kmer, k.data(), k.size() * 8);
        if (score) memcpy(score, sc.data(), sc.size() * 8);
        if (row) memcpy(row, r.data(), r.size() * 8);
    });
}
int kgwas_heap_output_list(const kgwas_heap* h, uint64_t* kmer, uint64_t* rank, uint64_t* row) {
    return guarded([&] {
        if (!h || !kmer || !rank || !row) throw Error(KGWAS_ERR_ARG, "kgwas_heap_output_list: null");
        std::vector<uint64_t> k, r;
        std::vector<double> sc;
        h->h.pop_all(k, sc, r);
        const size_t n = k.size();
        std::vector<size_t> idx(n);
        for (size_t i = 0; i < n; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return r[a] < r[b]; });
        for (size_t i = 0; i < n; i++) {
            kmer[i] = k[idx[i]];
            rank[i] = n - idx[i];
            row[i] = r[idx[i]];
        }
    });
}
int kgwas_heap_rows_sorted(const kgwas_heap* h, uint64_t* row) {
    return guarded([&] {
        if (!h || !row) throw Error(KGWAS_ERR_ARG, "kgwas_heap_rows_sorted: null");
        const std::vector<uint64_t> r = h->h.rows_sorted();
        std::copy(r.begin(), r.end(), row);
    });
}
int kgwas_heap_output_to_file(const kgwas_heap* h, const char* path, int with_scores) {
    return guarded([&] {
        if (!h || !path) throw Error(KGWAS_ERR_ARG, "kgwas_heap_output_to_file: null");
        std::vector<uint64_t> k, r;
        std::vector<double> sc;
        h->h.pop_all(k, sc, r);
        FILE* f = fopen(path, "wb");
        if (!f) throw Error(KGWAS_ERR_IO, std::string("cannot open ") + path);
        bool ok = true;
        for (size_t i = 0; i < k.size() && ok; i++) {
            ok = fwrite(&k[i], 8, 1, f) == 1;
            if (ok && with_scores) ok = fwrite(&sc[i], 8, 1, f) == 1;
        }
        ok = (fclose(f) == 0) && ok;
        if (!ok) throw Error(KGWAS_ERR_IO, std::string("cannot write ") + path);
    });
}
void kgwas_heap_free(kgwas_heap* h) { delete h; }

int kgwas_merge_shards(uint64_t n_pheno, const uint64_t* topn, uint64_t n_shards, const uint64_t* counts,
                       const uint64_t* const* kmer, const double* const* score, const uint64_t* const* row,
                       uint32_t threads, kgwas_heap** out_heaps) {
    return guarded([&] {
        if (!topn || !counts || !kmer || !score || !row || !out_heaps) throw Error(KGWAS_ERR_ARG, "kgwas_merge_shards: null");
        for (uint64_t j = 0; j < n_pheno; j++) {
            if (topn[j] == 0) throw Error(KGWAS_ERR_ARG, "heap size must be >= 1");
            out_heaps[j] = new kgwas_heap((size_t)topn[j]);
        }
        std::vector<std::vector<uint64_t>> off(n_shards, std::vector<uint64_t>(n_pheno + 1, 0));
        for (uint64_t g = 0; g < n_shards; g++)
            for (uint64_t j = 0; j < n_pheno; j++) off[g][j + 1] = off[g][j] + counts[g * n_pheno + j];
        unsigned nt = threads ? threads : usable_cpus();
        nt = (unsigned)std::min<uint64_t>(nt, n_pheno);
        Pool pool(nt);
        pool.parallel_for(n_pheno, [&](size_t j) {
            BestHeap& h = out_heaps[j]->h;
            for (uint64_t g = 0; g < n_shards; g++) {  // shards in row order
                const uint64_t o = off[g][j], n = counts[g * n_pheno + j];
                for (uint64_t i = 0; i < n; i++) h.add(kmer[g][o + i], score[g][o + i], (size_t)row[g][o + i]);
            }
        });
    });
}

}  // extern "C"
