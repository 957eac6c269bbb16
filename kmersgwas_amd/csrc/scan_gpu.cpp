// scan_gpu.cpp — the GPU side of a chunk of the association scan: kernel arguments, the dense (heap-filling) chunks, the
// sparse chunks (filter -> survivors' bitmap -> keys -> exact re-score -> compaction -> threshold update), their counts
// and record copies. The host replay of what they ship is scan_replay.cpp.
#include "scan_internal.h"

namespace kgwas {


void fill_args(kgwas_scan* s, ScoreArgs& a, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row,
               bool squeezed) {
    memset(&a, 0, sizeof(a));
    if (squeezed) {
        a.src.base = s->d_sq.p;
        a.src.stride_dw = 2 * s->W_m;
        a.src.off_dw = 0;
        a.src.avail_dw = (uint32_t)(2 * s->W_m);
    } else {
        a.src.base = reinterpret_cast<const uint32_t*>(d_rows);
        a.src.stride_dw = 2 * (1 + s->W_f);
        a.src.off_dw = 2;
        a.src.avail_dw = (uint32_t)(2 * s->W_f);
    }
    a.dmask = s->d_dmask.p;
    a.file_rows = d_rows;
    a.file_stride_w = 1 + s->W_f;
    a.n_rows = n_rows;
    a.first_row = first_row;
    a.S = (uint32_t)s->S;
    a.W_m = (uint32_t)s->W_m;
    a.n_pheno = (uint32_t)s->n_pheno;
    a.min_count = (uint32_t)std::min<uint64_t>(s->min_count, 0xFFFFFFFFull);
    a.Yperm = s->d_Yperm.p;
    a.Ymfma = s->d_Ymfma.p;
    a.sums = s->d_sums.p;
    a.thr = s->d_thr.p;
}

// One block per row block (it walks every column-tile group itself): keep at least ~8 rounds of
// blocks over the 256 CUs so the last round's imbalance stays small; the launcher rounds up to
// the rows one pass of the block's waves covers.
uint32_t pick_rows_per_block(uint64_t n_rows, uint64_t /*n_ctiles*/) {
    // Bigger row blocks amortise the per-group LDS refills and barriers (probe: 78.0 / 79.2 / 80.3 % of
    // peak at 1024 / 2048 / 4096 rows per block on 4 M rows x 96 columns).
    for (uint32_t rpb : {4096u, 2048u, 1024u, 512u}) {
        if ((n_rows + rpb - 1) / rpb >= 2048) return rpb;
    }
    return 256u;
}

void launch_score(kgwas_scan* s, const ScoreArgs& a) {
    if (s->kernel_used == KGWAS_KERNEL_MFMA) {
        const uint64_t nct = (s->n_pheno + 15) / 16;
        KGWAS_HIP(launch_score_mfma(a, pick_rows_per_block(a.n_rows, nct), s->nb_full, s->stream));
    } else {
        KGWAS_HIP(launch_score_valu(a, s->stream));
    }
    s->st.score_launches++;
}

void maybe_squeeze(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows) {
    if (s->direct) return;
    KGWAS_HIP(launch_squeeze(d_rows, 1 + s->W_f, n_rows, s->d_colmap.p, (uint32_t)s->W_m, (uint32_t)s->W_f, s->d_sq.p,
                             s->stream));
}

// --pattern_counter: hash the presence/absence pattern of every MAC-passing row of this feed
// (update_presence_absence_pattern_counter, src/kmers_multiple_databases.cpp:376-380). A separate,
// bandwidth-bound pass over the fed rows; the distinct count is taken at finish.
void hash_patterns(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows) {
    const uint64_t need = s->pat_upper + n_rows;
    if (need > s->d_pat.n) {  // grow (amortised doubling); keep what is there
        DevBuf<uint64_t> bigger;
        bigger.alloc(std::max<uint64_t>(need, 2 * s->d_pat.n));
        if (s->d_pat.n) {
            KGWAS_HIP(hipMemcpyAsync(bigger.p, s->d_pat.p, s->d_pat.n * 8, hipMemcpyDeviceToDevice, s->stream));
            KGWAS_HIP(hipStreamSynchronize(s->stream));
        }
        std::swap(bigger.p, s->d_pat.p);
        std::swap(bigger.n, s->d_pat.n);
    }
    const uint64_t stride = 1 + s->W_f;
    for (uint64_t pos = 0; pos < n_rows; pos += s->chunk_max) {
        const uint64_t c = std::min<uint64_t>(s->chunk_max, n_rows - pos);
        ScoreArgs a;
        fill_args(s, a, d_rows + pos * stride, c, 0, !s->direct);
        maybe_squeeze(s, d_rows + pos * stride, c);
        KGWAS_HIP(launch_pattern_hash(a.src, s->d_dmask.p, c, (uint32_t)s->S, (uint32_t)s->W_m, a.min_count, s->d_pat.p,
                                      s->d_pat_cnt.p, s->stream));
    }
    s->pat_upper += n_rows;
}

// The exact heap minima as far as the host has replayed. They go to thr_host, which thr_update_kernel
// folds into the device's own thresholds; before the sparse phase starts they are the thresholds.
void upload_thresholds(kgwas_scan* s) {
    double* h = s->h_thr.p + (s->thr_flip % 8u) * s->n_pheno;  // ring: uploads may queue behind long kernels
    s->thr_flip++;
    // (a heap that is still filling has no bound to offer: its smallest entry so far may well exceed its final minimum)
    // (a column in select mode offers its pool's bound: racy reads of monotone values, as the heaps' minima are)
    for (uint64_t j = 0; j < s->n_pheno; j++) h[j] = s->lazy[j].on ? s->lazy[j].bound() : (s->heaps[j].full() ? s->heaps[j].lowest() : 0.0);
    KGWAS_HIP(hipMemcpyAsync(s->d_thr_host.p, h, s->n_pheno * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (!s->hist_ready)
        KGWAS_HIP(hipMemcpyAsync(s->d_thr.p, h, s->n_pheno * sizeof(double), hipMemcpyHostToDevice, s->stream));
}

// First sparse chunk: histogram bin 0 of every column starts at its current exact minimum.
void start_histograms(kgwas_scan* s) {
    for (uint64_t j = 0; j < s->n_pheno; j++) {
        const double low = s->sel_valid ? s->h_sel.p[j] : (s->lazy[j].on ? s->lazy[j].bound() : s->heaps[j].lowest());
        uint64_t bits = 0;
        if (low == low && low > 0) memcpy(&bits, &low, 8);
        s->h_hist_base.p[j] = (uint32_t)(bits >> HIST_SHIFT);
    }
    KGWAS_HIP(hipMemcpyAsync(s->d_hist_base.p, s->h_hist_base.p, s->n_pheno * sizeof(uint32_t), hipMemcpyHostToDevice,
                             s->stream));
    KGWAS_HIP(hipMemsetAsync(s->d_hist.p, 0, s->n_pheno * (size_t)HIST_BINS * sizeof(uint32_t), s->stream));
    s->hist_ready = true;
}

void refresh_full(kgwas_scan* s) {
    bool all = true;
    for (uint64_t j = 0; j < s->n_pheno; j++) all = all && (s->lazy[j].on ? s->lazy[j].full() : s->heaps[j].full());
    s->all_full = all;
}

// Dense chunk: every score comes back; replay every kept row into every heap.
// select: also pick each column's topn-th largest score of the chunk on the device and make it the device's threshold
// (d_thr, d_thr_host) - see feed_device_impl; h_sel / h_sel_info arrive with the scores.
void run_dense(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row, double* out_scores,
               uint32_t* out_n1, bool replay, bool select) {
    ScoreArgs a;
    auto td0 = std::chrono::steady_clock::now();
    hipEvent_t e0 = s->ev_d0, e1 = s->ev_d1, es = s->ev_ds;
    fill_args(s, a, d_rows, n_rows, first_row, !s->direct);
    a.dense = s->d_dense.p;
    a.n1_out = s->d_n1.p;
    a.kmer_out = s->d_kmer.p;
    a.tested = s->d_tested_dense.p;
    double lap_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // KGWAS_TRACE: host time of the calls below (a fresh process: first uses)
    int lap_i = 0;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&] {
        const auto n = std::chrono::steady_clock::now();
        if (lap_i < 8) lap_ms[lap_i++] = std::chrono::duration<double, std::milli>(n - lap_t).count();
        lap_t = n;
    };
    KGWAS_HIP(hipMemsetAsync(s->d_tested_dense.p, 0, TESTED_SHARDS * sizeof(unsigned long long), s->stream));
    KGWAS_HIP(hipEventRecord(es, s->stream));
    lap();  // 0: fill + event
    maybe_squeeze(s, d_rows, n_rows);
    lap();  // 1: squeeze
    KGWAS_HIP(hipEventRecord(e0, s->stream));
    launch_score(s, a);
    lap();  // 2: scorer
    KGWAS_HIP(hipEventRecord(e1, s->stream));
    if (select) {
        KGWAS_HIP(launch_dense_select(s->d_dense.p, s->d_n1.p, (uint32_t)n_rows, (uint32_t)s->n_pheno, (uint32_t)s->S,
                                      (uint32_t)std::min<uint64_t>(s->min_count, 0xFFFFFFFFull), s->d_topn.p, s->d_thr.p, s->d_thr_host.p,
                                      s->d_sel.p, s->d_sel_info.p, s->stream));
        KGWAS_HIP(hipMemcpyAsync(s->h_sel.p, s->d_sel.p, s->n_pheno * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        KGWAS_HIP(hipMemcpyAsync(s->h_sel_info.p, s->d_sel_info.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    }
    lap();  // 3: select + its copies
    // (by kernels into the mapped buffers: no SDMA device -> host transfer anywhere in a scan, kernels.h launch_copy_to_host)
    // A dense START's scores (9.5 MB at 101 columns: 0.17 ms of PCIe) travel on the copy stream, behind the scorer and beside the
    // selection and the first sparse chunk, which the caller submits before it waits for them (wait_dense_copy): the stream's
    // first sparse launch does not queue up behind a kernel that is parked on PCIe stores.
    static const bool dcopy_side = !exp_set("KGWAS_DENSE_COPY_INLINE");  // experiments: the copy on the scan's stream, as before
    if (select && dcopy_side && s->copy_stream) {
        // (an SDMA transfer, not the copying kernel: its waves, parked on PCIe stores, doubled the selection's time beside them -
        // 199 against 101 us under the profiler, profiles/r06_chunk_timeline.txt of the first version; tools/probe_d2h.hip: a
        // transfer costs the kernels beside it nothing)
        KGWAS_HIP(hipStreamWaitEvent(s->copy_stream, e1, 0));
        KGWAS_HIP(hipMemcpyAsync(s->h_dense.p, s->d_dense.p, s->n_pheno * n_rows * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        KGWAS_HIP(hipEventRecord(s->ev_dcopy, s->copy_stream));
        s->dense_copy_pending = true;
    } else
        KGWAS_HIP(launch_copy_to_host(s->d_dense.p, s->h_dense_dev, s->n_pheno * n_rows * sizeof(double), s->stream));
    lap();  // 4: scores' copy
    KGWAS_HIP(launch_copy_to_host(s->d_n1.p, s->h_n1_dev, n_rows * sizeof(uint32_t), s->stream));
    KGWAS_HIP(launch_copy_to_host(s->d_kmer.p, s->h_kmer_dev, n_rows * sizeof(uint64_t), s->stream));
    lap();  // 5: n1 + kmer copies
    const auto td1 = std::chrono::steady_clock::now();
    KGWAS_HIP(hipStreamSynchronize(s->stream));
    float ms = 0;
    KGWAS_HIP(hipEventElapsedTime(&ms, e0, e1));
    s->st.score_kernel_ms += ms;
    if (s->trace) {
        float pre = 0;
        KGWAS_HIP(hipEventElapsedTime(&pre, es, e0));
        fprintf(stderr, "[kgwas] dense chunk on the device: %.3f ms to queue it, %.3f ms until the stream was idle; squeeze %.3f ms, scorer %.3f ms; host calls: fill %.2f squeeze %.2f scorer %.2f select %.2f scores' copy %.2f n1/kmer copies %.2f ms\n",
                std::chrono::duration<double, std::milli>(td1 - td0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td1).count(), pre, ms,
                lap_ms[0], lap_ms[1], lap_ms[2], lap_ms[3], lap_ms[4], lap_ms[5]);
    }
    if (!s->direct) {
        KGWAS_HIP(hipEventElapsedTime(&ms, es, e0));
        s->st.squeeze_kernel_ms += ms;
    }
    s->st.chunks++;
    if (out_scores || !select) wait_dense_copy(s);
    if (out_scores) memcpy(out_scores, s->h_dense.p, s->n_pheno * n_rows * sizeof(double));
    if (out_n1) memcpy(out_n1, s->h_n1.p, n_rows * sizeof(uint32_t));
    if (!replay || select) return;  // select: the caller pushes the rows (dense_fill) after submitting sparse chunks
    dense_fill(s, n_rows, first_row, td0);
}

// The dense start's scores, copied on the copy stream (run_dense), are in host memory when this returns.
void wait_dense_copy(kgwas_scan* s) {
    if (!s->dense_copy_pending) return;
    s->dense_copy_pending = false;
    KGWAS_HIP(hipEventSynchronize(s->ev_dcopy));
}

// The host half of a dense chunk: every MAC-passing row into every heap. meanwhile: run by the calling thread while the
// pool's workers push (the control thread submits the first sparse chunks there).
void dense_fill(kgwas_scan* s, uint64_t n_rows, uint64_t first_row, std::chrono::steady_clock::time_point td0,
                const std::function<void()>* meanwhile) {
    wait_dense_copy(s);  // (a caller that had something to submit meanwhile has waited already)
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t S = s->S, mc = s->min_count;
    uint64_t kept = 0;
    for (uint64_t r = 0; r < n_rows; r++) {
        const uint64_t n1 = s->h_n1.p[r];
        if (S >= mc && n1 >= mc && n1 <= S - mc) kept++;
    }
    s->st.rows_tested += kept;
    s->st.candidates += kept * s->n_pheno;
    std::atomic<uint64_t> pushes(0);
    const std::function<void(size_t)> push_column = [&](size_t j) {
        BestHeap& h = s->heaps[j];
        const double* sc = s->h_dense.p + j * n_rows;
        uint64_t local = 0;
        if (s->lazy[j].on) {
            // select mode (scan_lazy.cpp): the rows go to the column's log and pool. A tie among the column's N largest so far
            // (a table whose rows repeat presence/absence patterns shows them at once) means its result will depend on the
            // heap's layout: the column is replayed from here on, beginning with what it has logged.
            LazyCol& L = s->lazy[j];
            for (uint64_t r = 0; r < n_rows; r++) {
                const uint64_t n1 = s->h_n1.p[r];
                if (!(S >= mc && n1 >= mc && n1 <= S - mc)) continue;
                L.add(s->h_kmer.p[r], sc[r], first_row + r);
            }
            if (!s->lazy_log_mode && L.ties_now()) local = lazy_materialize(s, j);
            pushes += local;
            return;
        }
        for (uint64_t r = 0; r < n_rows; r++) {
            const uint64_t n1 = s->h_n1.p[r];
            if (!(S >= mc && n1 >= mc && n1 <= S - mc)) continue;
            if (h.add(s->h_kmer.p[r], sc[r], (size_t)(first_row + r))) {
                local++;
                if (s->record_history) s->hist[j].push(s->h_kmer.p[r], sc[r], first_row + r);
            }
        }
        if (s->record_history) _mm_sfence();  // streaming stores of the history log
        pushes += local;
    };
    s->pool->start(s->n_pheno, push_column);
    if (meanwhile) {
        // push_column and what it captures live on this frame: whatever the callback throws (a HIP error while it submits
        // sparse chunks), the workers are waited for before the frame unwinds
        try {
            (*meanwhile)();
        } catch (...) {
            s->pool->wait(false);
            throw;
        }
    }
    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] dense fill: control thread done, fill %s\n", s->t_ms(), s->pool->finished() ? "done" : "running");
    s->pool->wait();
    s->pool->rethrow();  // (a column whose push log could not grow)
    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] dense fill done\n", s->t_ms());
    s->st.heap_pushes += pushes.load();
    s->st.replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (s->trace)
        fprintf(stderr, "[kgwas] dense chunk rows=%llu: device part %.3f ms, host fill %.3f ms\n", (unsigned long long)n_rows,
                std::chrono::duration<double, std::milli>(t0 - td0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    s->rows_done += n_rows;
    s->rows_submitted = std::max(s->rows_submitted, s->rows_done);
    refresh_full(s);
    upload_thresholds(s);
    s->st.dense_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count();
}

// Which operand set the next sparse chunk uses. The one-slice filter does half (or less) of the matrix work per row
// but lists several survivors per candidate, and every survivor costs an exact re-score (S lane-ops x 4); the
// two-slice filter lists ~1.1. Per row: tile-slices x S x 0.014 ps of filter against survivors x S x 0.15 ps of
// re-score (both measured at 1135 x 101 and 2048 x 201), so one slice wins once
//     candidates per row x (infl[0] - infl[1]) < mode_k x (tile-slices[1] - tile-slices[0]),  mode_k ~ 0.09,
// i.e. early in a scan (low thresholds, many candidates per row) the chunks take two slices, later one. infl[] are the
// survivors per candidate the finished chunks of each mode reported (fetch_records).
int pick_coarse_mode(const kgwas_scan* s) {
    if (!s->cmode[0].ready) return 1;
    if (!s->cmode[1].ready) return 0;
    const double cand_row = (double)s->sum_topn / (double)std::max<uint64_t>(s->rows_submitted, 1);
    auto eq = [](const kgwas_scan::CoarseMode& M) { return M.tile_slices_eq > 0 ? M.tile_slices_eq : (double)M.tile_slices; };
    const double tiles0 = eq(s->cmode[0]), tiles1 = eq(s->cmode[1]);
    return cand_row * std::max(0.0, s->infl_obs[0] - s->infl_obs[1]) < s->mode_k * (tiles1 - tiles0) ? 0 : 1;
}

// count_hist: first (and only) scoring of these rows in the sparse phase -> their candidates feed the
// device-side threshold histograms. Overflow re-runs must not count the same rows twice.
void submit_sparse(kgwas_scan* s, Slot& sl, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row,
                   bool count_hist) {
    ScoreArgs a;
    fill_args(s, a, d_rows, n_rows, first_row, !s->direct);
    a.cand = sl.d_cand;
    a.cand_cnt = sl.d_cnt.p;
    a.cap = s->cap;
    a.tested = sl.d_tested.p;
    if (count_hist) {
        if (!s->hist_ready) start_histograms(s);
        a.hist = s->d_hist.p;
        a.hist_base = s->d_hist_base.p;
        a.hist_bins = HIST_BINS;
    } else {
        // Overflow re-run of rows the device has ALREADY counted in its histograms: d_thr may by now
        // reflect these very rows (or later ones), which is only valid for rows after them. The re-run is
        // synchronous and in order, so the host heaps hold exactly the rows before this range: use their
        // minima, nothing newer.
        for (uint64_t j = 0; j < s->n_pheno; j++) s->h_thr_redo.p[j] = s->lazy[j].on ? s->lazy[j].bound() : s->heaps[j].lowest();
        KGWAS_HIP(hipMemcpyAsync(s->d_thr_redo.p, s->h_thr_redo.p, s->n_pheno * sizeof(double), hipMemcpyHostToDevice,
                                 s->stream));
        a.thr = s->d_thr_redo.p;
    }
    const bool use_coarse = s->coarse && count_hist;
    // The chunk's four events - start (ev_k0), filter done (ev_mid), kernels done (ev_k1), counts on the host (ev_counts) - are bound
    // to the prep launch, the last filter launch, the compaction and the tail launch (launch.h: a recorded event costs the stream
    // 2.9 us, a bound one nothing); ev_k0 is the END of the prep launch, i.e. the kernel statistics no longer count its 2-3 us.
    if (!s->direct) {
        KGWAS_HIP(hipEventRecord(sl.ev_sq0, s->stream));  // (only read when rows are squeezed)
        maybe_squeeze(s, d_rows, n_rows);
        KGWAS_HIP(hipEventRecord(sl.ev_k0, s->stream));
    }
    {
        // One launch for the chunk's counters, its survivors' bitmap ([column][64-row word], which a popcount scan turns
        // into row-ordered keys per column: no key list, no sort) and the narrow filter's per-segment counts - inside the
        // interval the statistics count as kernel time.
        const uint64_t n_words = (n_rows + 63) / 64;
        const uint32_t n_segs = (uint32_t)((n_words + 1023) / 1024);
        // (narrow scans: the thresholds are raised here, by the first blocks of this launch, from everything the chunks
        // before this one counted - not by a launch of its own behind each chunk)
        PrepThr tu;
        memset(&tu, 0, sizeof(tu));
        if (use_coarse && s->narrow && s->hist_ready) {
            tu.hist = s->d_hist.p;
            tu.hist_base = s->d_hist_base.p;
            tu.bins = HIST_BINS;
            tu.topn = s->d_topn.p;
            tu.thr_host = s->d_thr_host.p;
            tu.thr = s->d_thr.p;
        }
        // (wide path: the key launches clear the words they list, so the bitmap is zeroed here only before a session's first
        // chunk or after a chunk whose key launches were never queued - in full, the chunks' geometries differ)
        const bool zero_bitmap = use_coarse && (s->narrow || !s->bitmap_clean);
        const uint64_t zero_words = !zero_bitmap ? 0 : (s->narrow ? s->n_pheno * n_words : s->d_bitmap.n);
        const auto prep = [&] {
            return launch_chunk_prep(sl.d_cnt.p, (uint32_t)s->n_pheno, sl.d_tested.p, use_coarse ? s->d_key_count.p : nullptr,
                                     zero_bitmap ? s->d_bitmap.p : nullptr, zero_words,
                                     (use_coarse && s->narrow) ? s->d_bm_blocks.p : nullptr, (use_coarse && s->narrow) ? (uint32_t)s->n_pheno * n_segs : 0u,
                                     tu, s->stream);
        };
        if (s->direct) KGWAS_HIP(bind_stop(sl.ev_k0, s->stream, prep));
        else KGWAS_HIP(prep());
    }
    sl.used_coarse = use_coarse;
    if (use_coarse) {
        CoarseArgs c;
        memset(&c, 0, sizeof(c));
        c.src = a.src;
        c.n_rows = n_rows;
        c.S = a.S;
        c.n_pheno = a.n_pheno;
        c.min_count = a.min_count;
        c.n_kgroups = s->n_kgroups;
        const int cm = s->narrow ? 0 : pick_coarse_mode(s);
        const kgwas_scan::CoarseMode& M = s->cmode[cm];
        sl.coarse_mode = cm;
        sl.cand_est = (double)s->sum_topn * (double)n_rows / (double)std::max<uint64_t>(s->rows_submitted, 1);
        c.n_slices = M.slices;
        c.eg_max = M.eg_max;
        c.rall_max = M.rall_max;
        c.rmax_max = M.rmax_max;
        c.thr = a.thr;
        c.tested = a.tested;
        static const uint32_t rpb_env = (uint64_t)exp_int("KGWAS_COARSE_RPB", 0u);  // experiments
        const uint64_t n_words = (n_rows + 63) / 64;
        if (s->narrow) {
            NarrowArgs na;
            memset(&na, 0, sizeof(na));
            na.src = a.src;
            na.n_rows = n_rows;
            na.S = a.S;
            na.n_pheno = a.n_pheno;
            na.min_count = a.min_count;
            na.n_kgroups = s->n_kgroups;
            na.Bn = s->d_Bn.p;
            na.cols = s->d_ncols.p;
            na.thr = a.thr;
            na.bitmap = s->d_bitmap.p;
            na.words_per_col = n_words;
            na.tested = a.tested;
            na.pack1 = s->narrow_pack1 ? 1u : 0u;
            na.seg_cnt = s->d_bm_blocks.p;
            na.n_segs = (uint32_t)((n_words + 1023) / 1024);
            // (rows of the same device buffer behind this chunk, when the rows are read in place: only a feed's last chunk needs
            // the narrow filter's second, tiny launch)
            na.slack_rows = s->direct ? s->slack_rows : 0;
            // (short blocks: three 4-wave blocks share a CU and a launch's block count is rarely a multiple of the
            // 768 block slots, so long blocks leave CUs idle at the end of every launch: 4096 rows per block measured
            // 3.4 ms per 100 M rows, 768 rows 3.0; the large chunks of a table that fills the HBM - 48-128 M rows - take
            // 1280: 26.3 ms per 1.2 G rows against 27.1 with 768, 26.5 with 1024 or 1536)
            KGWAS_HIP(bind_stop(sl.ev_mid, s->stream, [&] { return launch_narrow(na, rpb_env ? rpb_env : (n_rows >= (1u << 24) ? 1280u : n_rows >= (1u << 18) ? 768u : 256u), s->stream); }));
        } else {
            s->bitmap_clean = false;  // (until the key launches are queued behind the filter)
            c.bitmap = s->d_bitmap.p;
            c.words_per_col = n_words;
            // rows per block: the operand tiles (up to 128 KB) are loaded into LDS once per block, so blocks are long
            // where the launch still fills the chip four times over
            for (uint32_t pi = 0; pi < M.n_parts; pi++) {
                const kgwas_scan::CoarsePart& Pt = M.part[pi];
                c.n_lgroups = Pt.n_lgroups;
                c.Bq = Pt.d_Bq.p;
                c.cols = Pt.d_cols.p;
                c.tested = pi == 0 ? a.tested : nullptr;  // every launch sees every row: one of them counts
                const bool last_part = pi + 1u == M.n_parts;
                const uint32_t rpb = rpb_env ? rpb_env : (n_rows >= (1u << 22) ? 4096u : n_rows >= (1u << 20) ? 2048u : 512u);
                if (M.mx) {
                    MxArgs x;
                    memset(&x, 0, sizeof(x));
                    x.src = c.src;
                    x.n_rows = n_rows;
                    x.S = c.S;
                    x.n_pheno = c.n_pheno;
                    x.min_count = c.min_count;
                    x.n_full = M.mx_full;
                    x.n_quarter = M.mx_quarter;
                    x.n_lgroups = Pt.n_lgroups;
                    x.n_slices = M.slices;
                    x.s1_fp6 = M.mx_s1_fp6;
                    x.scale0 = M.mx_scale0;
                    x.Bq = reinterpret_cast<const uint8_t*>(Pt.d_Bq.p);
                    x.cols = Pt.d_cols.p;
                    x.thr = c.thr;
                    x.bitmap = c.bitmap;
                    x.words_per_col = c.words_per_col;
                    x.tested = c.tested;
                    x.eg_max = c.eg_max;
                    x.rall_max = c.rall_max;
                    x.rmax_max = c.rmax_max;
                    const auto filt = [&] {
                        return Pt.stream ? launch_mxs(x, Pt.T, Pt.ng, Pt.stream - 1u, rpb, s->stream) : launch_mx(x, Pt.T, rpb, s->stream);
                    };
                    if (last_part) KGWAS_HIP(bind_stop(sl.ev_mid, s->stream, filt));
                    else KGWAS_HIP(filt());
                } else {
                    const auto filt = [&] { return launch_coarse(c, Pt.T, rpb, s->stream); };
                    if (last_part) KGWAS_HIP(bind_stop(sl.ev_mid, s->stream, filt));
                    else KGWAS_HIP(filt());
                }
            }
            if (!M.n_parts) KGWAS_HIP(hipEventRecord(sl.ev_mid, s->stream));
        }
        a.tested = nullptr;  // counted by the filter
        a.so_score = sl.d_so_score.p;
        a.so_kmer = sl.d_so_kmer.p;
        a.so_row = sl.d_so_row.p;
        if (s->narrow) {
            // one to four columns: keys in one launch (the filter has counted its survivors per segment), every survivor's
            // record written in place by the re-score kernel - a chunk is five launches, not thirteen
            KGWAS_HIP(launch_narrow_keys(s->d_bitmap.p, n_words, n_rows, (uint32_t)s->n_pheno, s->d_bm_blocks.p, s->d_surv_sorted.p, s->key_slots,
                                         s->row_key_bits, s->d_surv_off.p, s->d_surv_cnt.p, s->d_key_count.p, s->d_tile_pref.p, sl.d_meta.p, sl.d_tested.p, s->stream));
            KGWAS_HIP(bind_stop(sl.ev_k1, s->stream, [&] { return launch_rescore_direct(a, s->d_surv_sorted.p, s->d_surv_off.p, s->d_surv_cnt.p, s->row_key_bits, s->d_tile_pref.p, s->stream); }));
        } else {
        KGWAS_HIP(launch_bitmap_keys(s->d_bitmap.p, n_words, n_rows, (uint32_t)s->n_pheno, s->d_bm_blocks.p, s->d_bm_mask.p, s->d_surv_sorted.p, s->key_slots,
                                     s->row_key_bits, s->d_surv_off.p, s->d_surv_cnt.p, s->d_key_count.p, s->d_tile_pref.p, /*nibble_transposed=*/!s->narrow, s->stream));
        s->bitmap_clean = true;
        a.so_score = sl.d_so_score.p;
        a.so_kmer = sl.d_so_kmer.p;
        a.so_row = sl.d_so_row.p;
        KGWAS_HIP(bind_stop(sl.ev_k1, s->stream, [&] {
            return launch_rescore(a, s->d_surv_sorted.p, s->d_surv_off.p, s->d_surv_cnt.p, s->row_key_bits, s->d_tile_pref.p,
                                  s->d_tile_cnt.p, s->d_tmp_score.p, s->d_key_count.p, sl.d_meta.p, s->stream);
        }));
        }
        s->st.score_launches++;
    } else {
        if (s->direct) KGWAS_HIP(hipEventRecord(sl.ev_k0, s->stream));  // (bound to nothing above: the prep launch of an exact-scorer chunk is not timed)
        launch_score(s, a);
        KGWAS_HIP(hipEventRecord(sl.ev_k1, s->stream));
    }
    const bool fused_tail = use_coarse && s->narrow;  // thresholds: raised by the next chunk's prep launch; the tested count: in meta (launch_narrow_keys)
    static const bool tail_kernel = !(exp_int("KGWAS_TAIL_KERNEL", 1) == 0);  // experiments: 0 = hipMemcpyAsync calls
    const bool one_launch = use_coarse && tail_kernel && sl.h_meta_dev;
    // wide path: the threshold update and the chunk's tail (counts, tested-row shards, thresholds into the mapped buffers) are one launch
    const bool thr_and_tail = one_launch && s->hist_ready && !fused_tail;
    if (s->hist_ready && !fused_tail && !thr_and_tail)  // raise the thresholds for whatever is queued next; no host round trip
        KGWAS_HIP(launch_thr_update(s->d_hist.p, s->d_hist_base.p, HIST_BINS, s->d_topn.p, s->d_thr_host.p, s->d_thr.p,
                                    (uint32_t)s->n_pheno, s->stream));
    if (!fused_tail && !one_launch)
        KGWAS_HIP(hipMemcpyAsync(sl.h_tested.p, sl.d_tested.p, TESTED_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                 s->stream));
    sl.tested_in_meta = fused_tail;
    if (thr_and_tail) {
        const bool thr_too = s->lazy_any.load(std::memory_order_relaxed);
        KGWAS_HIP(bind_stop(sl.ev_counts, s->stream, [&] {
            return launch_thr_tail(s->d_hist.p, s->d_hist_base.p, HIST_BINS, s->d_topn.p, s->d_thr_host.p, s->d_thr.p, (uint32_t)s->n_pheno, sl.d_meta.p,
                                   (uint32_t)(2 * s->n_pheno + 4), sl.h_meta_dev, sl.d_tested.p, (uint32_t)TESTED_SHARDS, sl.h_tested_dev,
                                   thr_too ? sl.h_thr_dev : nullptr, s->stream);
        }));
    } else if (one_launch) {
        // counts, tested-row shards and - for the columns in select mode, which bound their pools with them - the device's
        // thresholds as they stand behind this chunk: one launch into the mapped buffers; the record copies follow on the copy
        // stream once the control thread has read the counts (fetch_records)
        const bool thr_too = s->lazy_any.load(std::memory_order_relaxed);
        KGWAS_HIP(bind_stop(sl.ev_counts, s->stream, [&] {
            return launch_chunk_tail(sl.d_meta.p, (uint32_t)(2 * s->n_pheno + 4), sl.h_meta_dev, sl.d_tested.p, fused_tail ? 0u : (uint32_t)TESTED_SHARDS,
                                     sl.h_tested_dev, s->d_thr.p, thr_too ? (uint32_t)s->n_pheno : 0u, sl.h_thr_dev, s->stream);
        }));
    } else if (use_coarse) {
        // the record copies follow on the copy stream once the control thread has read the counts (fetch_records)
        KGWAS_HIP(hipMemcpyAsync(sl.h_meta.p, sl.d_meta.p, (2 * s->n_pheno + 4) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        // (columns in select mode prune their pools with the device's thresholds as they stand behind this chunk)
        if (s->lazy_any.load(std::memory_order_relaxed))
            KGWAS_HIP(hipMemcpyAsync(sl.h_thr.p, s->d_thr.p, s->n_pheno * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        KGWAS_HIP(hipEventRecord(sl.ev_counts, s->stream));
    } else {
        KGWAS_HIP(hipMemcpyAsync(sl.h_cnt.p, sl.d_cnt.p, s->n_pheno * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        KGWAS_HIP(hipEventRecord(sl.ev_done, s->stream));
    }
    sl.rows = d_rows;
    sl.first_row = first_row;
    sl.n_rows = n_rows;
    sl.busy = true;
    s->st.chunks++;
}

// The GPU side of a finished chunk: kernel timings into the statistics; returns false if a list overflowed (the
// chunk must then be redone, nothing of it may be replayed).
bool chunk_complete(kgwas_scan* s, Slot& sl) {
    sl.busy = false;
    float ms = 0;
    KGWAS_HIP(hipEventElapsedTime(&ms, sl.ev_k0, sl.ev_k1));
    s->st.score_kernel_ms += ms;
    if (sl.used_coarse) {
        float mc = 0;
        KGWAS_HIP(hipEventElapsedTime(&mc, sl.ev_k0, sl.ev_mid));
        s->st.coarse_kernel_ms += mc;
        s->st.coarse_launches++;
        s->st.coarse_mode_ms[sl.coarse_mode] += mc;
        s->st.coarse_mode_launches[sl.coarse_mode]++;
        s->st.coarse_mode_rows[sl.coarse_mode] += sl.n_rows;
    }
    if (!s->direct) {
        float mq = 0;
        KGWAS_HIP(hipEventElapsedTime(&mq, sl.ev_sq0, sl.ev_k0));
        s->st.squeeze_kernel_ms += mq;
    }
    bool over = false;
    if (sl.used_coarse) {
        over = sl.h_meta.p[2 * s->n_pheno + 1] > s->key_slots;  // the survivor key list
    } else {
        for (uint64_t j = 0; j < s->n_pheno && !over; j++) over = sl.h_cnt.p[j] > s->cap;
    }
    if (s->trace) {
        uint64_t tot = 0, mx = 0;
        for (uint64_t q = 0; q < s->n_pheno; q++) {
            const uint64_t v = sl.used_coarse ? sl.h_meta.p[q] : sl.h_cnt.p[q];
            tot += v;
            mx = std::max<uint64_t>(mx, v);
        }
        fprintf(stderr, "[kgwas] chunk rows=%llu first=%llu kernel=%.3fms records %llu (max per column %llu)%s%s\n",
                (unsigned long long)sl.n_rows, (unsigned long long)sl.first_row, ms, (unsigned long long)tot,
                (unsigned long long)mx, sl.used_coarse ? (" survivors " + std::to_string(sl.h_meta.p[2 * s->n_pheno + 1])).c_str() : "",
                over ? " OVERFLOW" : "");
    }
    if (over) return false;
    if (sl.tested_in_meta)
        s->st.rows_tested += (uint64_t)sl.h_meta.p[2 * s->n_pheno + 2] | ((uint64_t)sl.h_meta.p[2 * s->n_pheno + 3] << 32);
    else
        for (uint32_t i = 0; i < TESTED_SHARDS; i++) s->st.rows_tested += sl.h_tested.p[i];
    return true;
}

uint64_t next_sparse_chunk(const kgwas_scan* s) {
    // The device keeps its thresholds current with everything submitted so far (thr_update_kernel), so a
    // chunk of c rows ships about topn * c / rows_submitted records per column. Exact scorer: each column's list
    // holds cap records, keep that under cap / 3. Int8 filters: the survivor keys of all columns share one list of
    // key_slots (and so do the records); plan for half of it with the survivors per candidate that the finished
    // chunks of the coming chunk's mode reported (~1 for the narrow filter).
    const double m = (double)std::max<uint64_t>(s->rows_submitted, 1);
    // (Scans of one to four columns, the narrow filter: an eighth of that. Their few records would allow chunks of a third
    // of the table, but the thresholds a chunk is filtered against are those of its start, and the single heap's replay
    // is the scan's critical path: 16 chunks instead of 7 halve the records the host has to look at and reject - 298 k
    // -> 147 k at 100 M rows x 1 column, replay 4.8 -> 4.4 ms.)
    static const double fill_env = exp_num("KGWAS_FILL", 0.0);  // experiments
    // (... as long as the column IS replayed: in select mode - scan_lazy.cpp - a record costs the host a copy and a compare, and
    // nine chunks instead of sixteen take the one-column pass over 100 M rows from 3.94 to 3.66 ms)
    const double fill = fill_env > 0.0 ? fill_env : (s->narrow ? (s->lazy_any.load(std::memory_order_relaxed) ? 0.15 : 0.05) : 0.4);
    double c;
    if (s->coarse) {
        const double infl = s->narrow ? 1.0 : std::max(1.0, s->infl_obs[pick_coarse_mode(s)]);
        c = fill * m * (double)s->key_slots / (infl * (double)std::max<uint64_t>(s->sum_topn, 1));
    } else {
        c = m * (double)s->cap / (3.0 * (double)std::max<uint64_t>(s->max_topn, 1));
    }
    uint64_t ci = (uint64_t)std::min<double>(c, (double)s->chunk_max);
    ci = std::max<uint64_t>(ci, std::min<uint64_t>(s->dense_rows, s->chunk_max));
    ci = std::min<uint64_t>(ci, s->chunk_max);
    return (ci + 127) / 128 * 128;
}

// Wait for an event of a sparse chunk: poll for a while (an event that is about to complete is seen within a
// microsecond or two), then sleep in the driver (hipEventBlockingSync: the control thread must not occupy a CPU beside
// the replay workers while the GPU works on a long chunk). A sleeping wait alone costs 50-500 us per wake-up, twice per
// chunk, which is what a scan with few columns and few chunks then consists of.
void wait_event(kgwas_scan* s, hipEvent_t ev) {
    auto w0 = std::chrono::steady_clock::now();
    bool done = false;
    static const int polls = (int)exp_int("KGWAS_WAIT_POLLS", 400);  // experiments
    for (int i = 0; i < polls && !done; i++) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) done = true;
        else if (q != hipErrorNotReady) KGWAS_HIP(q);
        else
            for (int k = 0; k < 20; k++) __builtin_ia32_pause();
    }
    if (!done) KGWAS_HIP(hipEventSynchronize(ev));
    s->st.gpu_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
}

// The ring has run full while the columns' logs refer to its records (ring_keep). Called by the control thread when every
// published chunk is replayed (no worker is inside a column): the logs take copies, and the ring is recycled from here on -
// free up to the end of the last replayed chunk; chunks that are fetched but not yet published keep their records where
// they are and are copied into the logs when their turn comes.
void ring_to_recycling(kgwas_scan* s, uint64_t replayed) {
    for (uint64_t j = 0; j < s->n_pheno; j++)
        if (s->lazy[j].on) s->lazy[j].detach();
    s->ring_keep.store(false, std::memory_order_release);
    s->ring_tail = replayed ? s->slot[(size_t)((replayed - 1) % (uint64_t)s->n_slots)].ring_end : s->ring_feed_start;
    s->ring_freed = replayed;
    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] record ring full: logs detached, ring recycles from %zu (head %zu of %zu)\n", s->t_ms(), s->ring_tail, s->ring_head, s->ring_size);
}

// Coarse chunk: wait for its counts, then order the copy of exactly that many candidate records (three arrays) from
// HBM on the copy stream; ev_done follows the copies. Other chunks recorded ev_done at submission.
// Returns false if the record ring has no room yet (nothing was ordered: retry after more chunks are replayed).
bool fetch_records(kgwas_scan* s, Slot& sl, uint64_t seq) {
    if (!sl.used_coarse) return true;
    wait_event(s, sl.ev_counts);
    const uint32_t n = sl.h_meta.p[2 * s->n_pheno];
    const uint32_t n_surv = sl.h_meta.p[2 * s->n_pheno + 1];
    const bool copy = n && n_surv <= s->key_slots;
    {
        // give back what the replay has finished with (chunks are replayed, hence freed, in order); before this slot's
        // ring_end is overwritten below: its previous chunk is among them
        const uint64_t rep = s->seq_replayed.load(std::memory_order_acquire);
        if (s->ring_keep.load(std::memory_order_relaxed)) s->ring_freed = rep;  // (the logs refer to the records: nothing is given back)
        while (s->ring_freed < rep) {
            s->ring_tail = s->slot[(size_t)(s->ring_freed % (uint64_t)s->n_slots)].ring_end;
            s->ring_freed++;
        }
        // empty: start over at the bottom - but only when every chunk fetched before this one has been freed: a fetched,
        // not yet replayed chunk without records (or one that overflowed) carries a ring_end taken from the old head,
        // and freeing it later would move the tail back over records placed at the bottom in the meantime
        if (!s->ring_keep.load(std::memory_order_relaxed) && s->ring_tail == s->ring_head && s->ring_freed == seq) s->ring_head = s->ring_tail = 0;
    }
    if (copy) {
        const size_t need = ((size_t)n * 20 + 63) / 64 * 64;
        size_t at;
        if (s->ring_keep.load(std::memory_order_relaxed)) {  // linear: [0, head) is referred to by the columns' logs
            if (s->ring_size - s->ring_head < need) return false;  // (the caller lets the replay catch up, then ring_to_recycling)
            at = s->ring_head;
        } else if (s->ring_head >= s->ring_tail) {  // used part does not wrap (or the ring is empty)
            if (s->ring_size - s->ring_head >= need)
                at = s->ring_head;
            else if (s->ring_tail > need)  // wrap: the bytes up to the end stay unused until this chunk is freed
                at = 0;
            else
                return false;
        } else {
            if (s->ring_tail - s->ring_head > need)
                at = s->ring_head;
            else
                return false;
        }
        sl.so_score = reinterpret_cast<double*>(s->ring.p + at);
        sl.so_kmer = reinterpret_cast<uint64_t*>(s->ring.p + at + (size_t)n * 8);
        sl.so_row = reinterpret_cast<uint32_t*>(s->ring.p + at + (size_t)n * 16);
        s->ring_head = at + need;
    }
    sl.ring_end = s->ring_head;
    if (!s->narrow && n_surv > s->key_slots)  // the list overflowed (the chunk is redone by the exact scorer): plan the next chunks for what it saw
        s->infl_obs[sl.coarse_mode] = std::min(256.0, std::max(s->infl_obs[sl.coarse_mode], 1.25 * (double)n_surv / std::max(sl.cand_est, 1.0)));
    else if (!s->narrow && n >= 1024)
        s->infl_obs[sl.coarse_mode] = std::min(64.0, std::max(1.0, (double)n_surv / (double)n));
    // Two ways to the host. Beside a streamed feed's 128 MiB host -> device pieces, hipMemcpyAsync device -> host transfers
    // complete ~2 ms late (every piece's consumer call took 3 ms instead of 0.55): there the GPU writes the records into the
    // mapped ring itself, one launch. Over a table resident in HBM the transfers are punctual and cost the compute units
    // nothing, while the copying kernel's waves - parked on PCIe stores - slow the filter beside them (10.2 -> 12.7 ms per
    // 100 M rows x 101 columns): there the three transfers stay. KGWAS_RECORD_COPY=kernel|memcpy forces one (experiments).
    static const char* rc_env = opt_str("KGWAS_RECORD_COPY");
    const bool by_memcpy = rc_env ? strcmp(rc_env, "memcpy") == 0 : !s->streamed_feed;
    if (copy && !by_memcpy) {
        uint8_t* dev_at = s->ring_dev + (reinterpret_cast<uint8_t*>(sl.so_score) - s->ring.p);
        KGWAS_HIP(launch_records_to_host(sl.d_so_score.p, sl.d_so_kmer.p, sl.d_so_row.p, n, reinterpret_cast<double*>(dev_at),
                                         reinterpret_cast<uint64_t*>(dev_at + (size_t)n * 8), reinterpret_cast<uint32_t*>(dev_at + (size_t)n * 16), s->copy_stream));
    } else if (copy) {
        KGWAS_HIP(hipMemcpyAsync(sl.so_score, sl.d_so_score.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        KGWAS_HIP(hipMemcpyAsync(sl.so_row, sl.d_so_row.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->copy_stream));
        KGWAS_HIP(hipMemcpyAsync(sl.so_kmer, sl.d_so_kmer.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, s->copy_stream));
    }
    KGWAS_HIP(hipEventRecord(sl.ev_done, s->copy_stream));
    return true;
}

}  // namespace kgwas
