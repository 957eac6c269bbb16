// scan_replay.cpp — the host side of the association scan: the replay of a chunk's candidate records through the heaps
// ((chunk, column group) units), the pool's streaming workers, overflow recovery, and the feed loop that submits chunks,
// orders their record copies, publishes them to the workers and waits for the replay.
#include "scan_internal.h"

namespace kgwas {

void process_range_sync(kgwas_scan* s, Slot& sl, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row);

// ---- replay of a sparse chunk's records -------------------------------------------------------------------
// The unit of host work is (chunk, column group): group g owns the columns g, g + n_groups, ... and replays a
// chunk's records of those columns in row order. A group's chunks must be replayed in submission order; different
// groups are independent (columns never interact). Two drivers use it: the streaming replay below (workers pick
// (chunk, group) units as the GPU completes chunks, no barrier between chunks) and the synchronous overflow path.
// KGWAS_TIE_CHECKS (experiments): 0 = the pools are never looked at for ties during a large feed, 1 = behind every flagged
// chunk; default: where the host keeps up with the GPU.
static int tie_checks_mode() {
    static const int m = (int)exp_int("KGWAS_TIE_CHECKS", -1);
    return m;
}

void replay_group(kgwas_scan* s, Slot& sl, size_t g, ReplayAcc& acc) {
    const auto tb0 = std::chrono::steady_clock::now();
    const uint64_t row0 = sl.first_row;
    const std::vector<uint32_t>& members = s->grp_cols[g];
    uint64_t local = 0, nc = 0;
    if (sl.used_coarse) {
        // Records arrive in row order (sorted on the device). The group's columns advance together: per round, every
        // column scans forward to its next record that beats the column's current minimum (everything else is a
        // no-op for add_association), then the heaps of equal size take their replacements in lockstep (heap.h).
        // Columns are independent, so interleaving them changes nothing in any column's own sequence of pushes.
        const double none = -std::numeric_limits<double>::infinity();
        struct Cur {
            const double* sc;
            const uint64_t* km;
            const uint32_t* rw;
            uint32_t i, n;
            BestHeap* h;
            size_t j;
        };
        constexpr int MK = BestHeap::MAX_LOCKSTEP;
        Cur cols[MK];
        size_t n_cols = 0;
        const bool by_ref = s->ring_keep.load(std::memory_order_acquire);  // (constant while chunks are being replayed)
        // A look at a column's pool costs ~0.1 ms, and the replay it may set off ~4 ms x the share of the feed that is over. A
        // worker with nothing else waiting has that time beside the GPU; one that has fallen behind - few threads for many
        // columns: the host is the scan's bottleneck whenever its replays happen - would only add the looks to its backlog
        // (two threads under the headline: 47 ms per step without looks, 59 with).
        const bool look = sl.tie_check && !s->lazy_log_mode &&
                          (tie_checks_mode() == 1 || s->seq_published.load(std::memory_order_relaxed) <= s->gstate[g].done.load(std::memory_order_relaxed) + 2);
        for (const uint32_t j : members) {
            const uint32_t n = sl.h_meta.p[j];
            if (!n) continue;
            const uint64_t o = sl.h_meta.p[s->n_pheno + j];
            if (s->lazy[j].on) {
                // a column in select mode (scan_lazy.cpp): its records go to its log and, where they can still be among the
                // N largest, to its pool - no heap is touched
                LazyCol& L = s->lazy[j];
                uint64_t tb;
                memcpy(&tb, &sl.h_thr.p[j], 8);
                L.take_chunk(sl.so_score + o, sl.so_kmer + o, sl.so_row + o, n, row0, tb, by_ref);
                nc += n;
                // behind chunks the control thread picked (flag_tie_check) the pool is looked at for a tie: a column whose N
                // largest scores are not distinct will need the exact replay, and that costs less now, beside the GPU, than at
                // finish
                if (look && L.full() && L.ties_now()) local += lazy_materialize(s, j);
                continue;
            }
            cols[n_cols++] = Cur{sl.so_score + o, sl.so_kmer + o, sl.so_row + o, 0, n, &s->heaps[j], j};
        }
        // The records were just written by the GPU (no CPU cache holds them) and the replay walks several short
        // streams at once, more than the hardware prefetchers track: pull them in up front, a line at a time.
        // (Prefetching ALL of a unit's records up front - up to a megabyte - pushed the heaps out of the L2 and the
        // records' own first lines out again before their turn: 10 % of the replay's CPU time.)
        static const uint32_t PF_AHEAD = (uint64_t)exp_int("KGWAS_REPLAY_PF", 64u);  // records
        auto pf = [&](const void* p) { __builtin_prefetch(p, 0, 3); };
        for (size_t c = 0; c < n_cols; c++) {
            const Cur& cu = cols[c];
            const uint32_t lim = std::min<uint32_t>(cu.n, PF_AHEAD + 8);
            for (uint32_t i = 0; i < lim; i += 8) pf(cu.sc + i);
            for (uint32_t i = 0; i < lim; i += 8) pf(cu.km + i);
            for (uint32_t i = 0; i < lim; i += 16) pf(cu.rw + i);
        }
        auto advance = [&](Cur& cu) {  // one record consumed; rolling prefetch a few lines ahead of the cursor
            cu.i++;
            if ((cu.i & 7u) == 0u && cu.i + PF_AHEAD < cu.n) {
                pf(cu.sc + cu.i + PF_AHEAD);
                pf(cu.km + cu.i + PF_AHEAD);
                if ((cu.i & 15u) == 0u) pf(cu.rw + cu.i + PF_AHEAD);
            }
        };
        const bool prof = s->trace;
        uint64_t q_scan = 0, q_heap = 0;
        while (n_cols) {
            const uint64_t q0 = prof ? __builtin_ia32_rdtsc() : 0;
            // next effective record of every column still active
            for (size_t c = 0; c < n_cols;) {
                Cur& cu = cols[c];
                bool ready = false;
                // (nothing touches this column's heap while its records are scanned: bound and fill state in registers)
                const bool open = !cu.h->full();
                const double low = cu.h->lowest();
                uint64_t rejected = 0;
                while (cu.i < cu.n) {
                    const double v = cu.sc[cu.i];
                    if (v == none) {  // a narrow chunk's survivor that is no candidate (launch_rescore_direct): never an add_association call
                        advance(cu);
                        continue;
                    }
                    if (open || v > low) {
                        ready = true;
                        break;
                    }
                    rejected++;
                    advance(cu);
                }
                nc += rejected;
                cu.h->note_rejected(rejected);
                if (ready)
                    c++;
                else
                    cols[c] = cols[--n_cols];
            }
            // lockstep groups of equal heap size (columns that differ, or are not full, go one at a time)
            const uint64_t q1 = prof ? __builtin_ia32_rdtsc() : 0;
            q_scan += q1 - q0;
            size_t done = 0;
            while (done < n_cols) {
                BestHeap* hp[MK];
                uint64_t km[MK], rw[MK];
                double sc[MK];
                Cur* who[MK];
                int K = 0;
                const size_t cap0 = cols[done].h->capacity();
                const bool full0 = cols[done].h->full();
                size_t c = done;
                for (; c < n_cols && K < MK; c++) {
                    Cur& cu = cols[c];
                    if (cu.h->capacity() != cap0 || !cu.h->full() || !full0) break;
                    hp[K] = cu.h;
                    km[K] = cu.km[cu.i];
                    sc[K] = cu.sc[cu.i];
                    rw[K] = row0 + cu.rw[cu.i];
                    who[K] = &cu;
                    K++;
                }
                if (K == 0) {  // not full: plain add_association
                    Cur& cu = cols[done];
                    hp[0] = cu.h;
                    km[0] = cu.km[cu.i];
                    sc[0] = cu.sc[cu.i];
                    rw[0] = row0 + cu.rw[cu.i];
                    who[0] = &cu;
                    cu.h->add(km[0], sc[0], (size_t)rw[0]);
                    K = 1;
                    c = done + 1;
                } else {
                    BestHeap::replace_top_n(K, hp, km, sc, rw);
                }
                for (int k = 0; k < K; k++) {
                    if (s->record_history) s->hist[who[k]->j].push(km[k], sc[k], rw[k]);
                    advance(*who[k]);
                }
                local += (uint64_t)K;
                nc += (uint64_t)K;
                done = c;
            }
            if (prof) q_heap += __builtin_ia32_rdtsc() - q1;
        }
        if (prof) {
            s->prof_scan.fetch_add(q_scan, std::memory_order_relaxed);
            s->prof_heap.fetch_add(q_heap, std::memory_order_relaxed);
        }
    } else {
        for (const uint32_t j : members) {
            const uint32_t n = sl.h_cnt.p[j];
            if (!n) continue;
            const Cand* c = sl.cand.p + j * (uint64_t)s->cap;
            BestHeap& h = s->heaps[j];
            // The device filtered against a minimum that is one chunk old. Anything not above the
            // CURRENT minimum would be rejected by add_association whenever it arrives (the minimum only
            // rises), so it is dropped before the sort. The survivors are put in row order through
            // compact (row-in-chunk, index) keys.
            std::vector<uint64_t>& keys = s->keys[j];
            keys.clear();
            LazyCol& L = s->lazy[j];
            const bool full = L.on ? false : h.full();
            const double low = h.lowest();
            for (uint32_t i = 0; i < n; i++)
                if (!full || c[i].score > low) keys.push_back(((c[i].row - row0) << 32) | i);
            std::sort(keys.begin(), keys.end());
            if (L.on) {  // select mode: everything that came, in row order, into the column's log
                for (uint64_t key : keys) {
                    const Cand& e = c[(uint32_t)key];
                    L.add(e.kmer, e.score, e.row);
                }
                nc += n;
                continue;
            }
            for (uint64_t key : keys) {
                const Cand& e = c[(uint32_t)key];
                if (h.add(e.kmer, e.score, (size_t)e.row)) {
                    local++;
                    if (s->record_history) s->hist[j].push(e.kmer, e.score, e.row);
                }
            }
            nc += n;
        }
    }
    if (s->record_history) _mm_sfence();  // streaming stores of the history log
    acc.pushes += local;
    acc.cands += nc;
    acc.units++;
    acc.busy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tb0).count();
}

void add_replay_stats(kgwas_scan* s, const ReplayAcc& a) {
    s->st.heap_pushes += a.pushes;
    s->st.candidates += a.cands;
    s->st.replay_cpu_ms += (double)a.busy_ns * 1e-6;
}

// Synchronous: wait for a submitted sparse chunk, replay its records (all groups, one barrier), refresh thresholds.
// Returns false if a candidate list overflowed (nothing was replayed). Overflow recovery only.
bool reap_sparse(kgwas_scan* s, Slot& sl) {
    {
        auto w0 = std::chrono::steady_clock::now();
        KGWAS_HIP(hipEventSynchronize(sl.ev_done));  // kernel done (mapped candidate writes visible) + counts copied
        s->st.gpu_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    }
    if (!chunk_complete(s, sl)) return false;
    auto t0 = std::chrono::steady_clock::now();
    const size_t NG = s->n_groups.load(std::memory_order_acquire);
    std::vector<ReplayAcc> accs(NG);
    s->pool->parallel_for(NG, [&](size_t g) { replay_group(s, sl, g, accs[g]); });
    for (const ReplayAcc& a : accs) add_replay_stats(s, a);
    s->st.replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s->rows_done += sl.n_rows;
    upload_thresholds(s);
    return true;
}

// Synchronous processing of a row range (overflow recovery): halve until the lists fit,
// fall back to dense chunks for very small ranges.
void process_range_sync(kgwas_scan* s, Slot& sl, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row) {
    const uint64_t stride = 1 + s->W_f;
    if (n_rows <= s->dense_rows) {
        run_dense(s, d_rows, n_rows, first_row, nullptr, nullptr, true);
        return;
    }
    Slot& rs = s->coarse ? s->redo : sl;  // coarse-mode slots carry no exact-scorer candidate buffers
    submit_sparse(s, rs, d_rows, n_rows, first_row, /*count_hist=*/false);
    if (reap_sparse(s, rs)) return;
    const uint64_t half = n_rows / 2;
    process_range_sync(s, sl, d_rows, half, first_row);
    process_range_sync(s, sl, d_rows + half * stride, n_rows - half, first_row + half);
}

// ---- streaming replay ----------------------------------------------------------------------------------------
// Sparse chunks carry sequence numbers (slot = seq % n_slots). The control thread (the caller of a feed) submits
// chunks while slots are free, waits for the GPU to finish them in order and PUBLISHES them (seq_published); the
// pool's workers, running one long item each, pick (chunk, group) units: any group that is not being worked on and
// whose next chunk is published, the group furthest behind first. There is no barrier between chunks, so a chunk's
// slowest group does not hold the others up (the per-chunk barrier cost 17 % of the replay at 101 columns on 16
// workers), and the pool is woken once per feed instead of once per chunk. A slot is reused when all groups have
// replayed its chunk (seq_replayed counts such chunks; they complete in order).
// A group that has fallen behind while other workers have run out of work (a co-tenant on its worker's CPU: the pool's
// threads are pinned, and a group's chunks can only be replayed one after the other) is cut into single-column groups
// that anybody takes: its backlog then runs on as many CPUs as it has columns instead of on the one slow CPU. Called by
// the worker that holds the group (busy), between two units; `done` = chunks the group has replayed.
static void split_group(kgwas_scan* s, size_t g, uint64_t done) {
    std::vector<uint32_t>& cols = s->grp_cols[g];
    if (cols.size() < 2) return;
    std::lock_guard<std::mutex> lk(s->split_mu);
    size_t n = s->n_groups.load(std::memory_order_relaxed);
    for (size_t i = 1; i < cols.size(); i++, n++) {
        s->grp_cols[n].assign(1, cols[i]);
        s->gstate[n].done.store(done, std::memory_order_relaxed);
        s->gstate[n].busy.store(0u, std::memory_order_relaxed);
        s->grp_owner[n].store(-1, std::memory_order_relaxed);
    }
    cols.resize(1);
    s->grp_owner[g].store(-1, std::memory_order_relaxed);
    s->n_groups.store(n, std::memory_order_release);  // (readers take n_groups with acquire: the entries above are complete)
    s->n_splits.fetch_add(1, std::memory_order_relaxed);
}

// kgwas_scan_expect_finish: a worker with nothing left to replay pops ONE column whose group has replayed every chunk of
// the feed (no push can follow) into its result lists, as kgwas_scan_finish would. One column per turn, claimed through
// col_popped (0 -> 1 in progress -> 2 done): the columns of the group that finishes last are then popped by as many idle
// workers as it has columns, instead of by its own worker alone after everybody else has long finished. Returns false
// if there is nothing left to claim.
static bool pop_ahead(kgwas_scan* s, size_t NG, uint64_t pub) {
    for (size_t g = 0; g < NG; g++) {
        if (s->gstate[g].done.load(std::memory_order_acquire) != pub) continue;
        for (const uint32_t j : s->grp_cols[g]) {
            uint8_t expect = 0;
            if (s->col_popped[j].load(std::memory_order_relaxed) != 0 ||
                !s->col_popped[j].compare_exchange_strong(expect, 1, std::memory_order_acq_rel))
                continue;
            lazy_finish_column(s, j);
            s->col_popped[j].store(2, std::memory_order_release);
            s->n_popped_ahead.fetch_add(1, std::memory_order_relaxed);
            return true;
        }
    }
    return false;
}

void replay_worker(kgwas_scan* s, size_t w) {
    ReplayAcc acc;
    int idle_spins = 0;
    bool hungry = false;  // counted in rp_hungry
    try {
        // The feed is over and kgwas_scan_finish comes next (kgwas_scan_expect_finish): the workers do not leave before every
        // column has its result lists - the columns that need their log replayed after all (a tie that turned up late: ~4 ms
        // each) side by side, the selections (0.1 ms each) around them. (Leaving as soon as the last chunk was replayed left
        // ONE such column to the worker that had started on it - the feed waited for it - and the others to finish: 4 + 5.5 ms
        // instead of 4.5.)
        auto drain = [&] {
            if (!s->rp_drain.load(std::memory_order_acquire) || s->rp_failed.load(std::memory_order_acquire)) return;
            while (pop_ahead(s, s->n_groups.load(std::memory_order_acquire), s->rp_final_pub.load(std::memory_order_relaxed))) {
            }
        };
        for (;;) {
            if (s->rp_quit.load(std::memory_order_acquire)) {
                drain();
                break;
            }
            const size_t NG = s->n_groups.load(std::memory_order_acquire);
            const uint64_t pub = s->seq_published.load(std::memory_order_acquire);
            // The group furthest behind among this worker's own groups and the floating ones; another worker's group
            // only when that worker has fallen two published chunks behind (a heap that changes cores drags its
            // 320 KB along, so groups stay at home unless a core is really slow - a co-tenant, a throttled sibling).
            size_t best = (size_t)-1;
            uint64_t best_done = ~0ull;
            for (int pass = 0; pass < 2 && best == (size_t)-1; pass++)
                for (size_t g = 0; g < NG; g++) {
                    const int home = s->grp_owner[g].load(std::memory_order_relaxed);
                    const bool mine = home < 0 || (size_t)home == w;
                    if (mine != (pass == 0)) continue;
                    kgwas_scan::GroupState& G = s->gstate[g];
                    if (G.busy.load(std::memory_order_relaxed)) continue;
                    const uint64_t d = G.done.load(std::memory_order_acquire);
                    if (d >= pub || (pass == 1 && d + 2 > pub)) continue;
                    if (d < best_done) {
                        best = g;
                        best_done = d;
                    }
                }
            if (best != (size_t)-1) {
                kgwas_scan::GroupState& G = s->gstate[best];
                uint32_t expect = 0;
                if (!G.busy.compare_exchange_strong(expect, 1u, std::memory_order_acq_rel)) continue;
                const uint64_t d = G.done.load(std::memory_order_acquire);
                if (d >= s->seq_published.load(std::memory_order_acquire)) {  // somebody else did it meanwhile
                    G.busy.store(0u, std::memory_order_release);
                    continue;
                }
                if (hungry) {
                    hungry = false;
                    s->rp_hungry.fetch_sub(1, std::memory_order_relaxed);
                }
                const size_t si = (size_t)(d % (uint64_t)s->n_slots);
                const double tr0 = s->trace ? s->t_ms() : 0.0;
                const uint32_t ncols = (uint32_t)s->grp_cols[best].size();  // (before a split below makes it 1)
                const auto tu0 = std::chrono::steady_clock::now();
                replay_group(s, s->slot[si], best, acc);
                if ((int)w == s->dbg_slow_worker) {  // experiments: this worker's CPU is shared with somebody else
                    const auto dur = std::chrono::steady_clock::now() - tu0;
                    const auto until = std::chrono::steady_clock::now() + std::max<std::chrono::steady_clock::duration>(dur * s->dbg_slow_pct / 100, std::chrono::microseconds(s->dbg_slow_min_us));
                    while (std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
                }
                if (s->trace && (NG <= 4 || best == 0))
                    fprintf(stderr, "[kgwas t=%.3f] worker %zu replayed chunk %llu group %zu in %.3f ms\n", s->t_ms(), w, (unsigned long long)d, best, s->t_ms() - tr0);
                G.done.store(d + 1, std::memory_order_release);
                // somebody has nothing to do and this group still owes two published chunks or more: cut it up
                // (only once the feed's last chunk is published: while the GPU still delivers, idle workers are waiting for
                // IT, and a group cut up then replays the rest of the feed as single columns at twice the cost per push)
                const uint64_t pub_now = s->seq_published.load(std::memory_order_acquire);
                if (s->split_lagging && ncols > 1 && s->rp_hungry.load(std::memory_order_relaxed) > 0 && pub_now >= d + 3 &&
                    s->rp_all_published.load(std::memory_order_acquire)) {
                    split_group(s, best, d + 1);
                    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] worker %zu split group %zu (%u columns) after chunk %llu\n", s->t_ms(), w, best, ncols, (unsigned long long)d);
                    s->rp_cv_work.notify_all();
                } else if (s->float_lead && s->grp_owner[best].load(std::memory_order_relaxed) == (int)w) {
                    // Nobody is idle yet, but this worker's own group is float_lead chunks behind the foremost group that
                    // is still at home: the group floats from now on - whole, so its heaps keep their lockstep - and the
                    // workers that are ahead take its chunks in turn ("the group furthest behind first"), as this one does
                    // at its own pace.
                    uint64_t lead = 0;
                    for (size_t g = 0; g < s->n_groups0; g++)
                        if (s->grp_owner[g].load(std::memory_order_relaxed) >= 0) lead = std::max(lead, s->gstate[g].done.load(std::memory_order_relaxed));
                    // (... and the foremost group itself has chunks waiting: the host is what the scan waits for)
                    if (lead >= d + 1 + s->float_lead && pub_now >= lead + 2) {
                        s->grp_owner[best].store(-1, std::memory_order_relaxed);
                        s->n_floated.fetch_add(1, std::memory_order_relaxed);
                        if (s->trace) fprintf(stderr, "[kgwas t=%.3f] worker %zu lets group %zu float after chunk %llu (%llu behind)\n", s->t_ms(), w, best, (unsigned long long)d, (unsigned long long)(lead - d - 1));
                    }
                }
                G.busy.store(0u, std::memory_order_release);
                idle_spins = 0;
                if (s->slot_left[si].fetch_sub(ncols, std::memory_order_acq_rel) == ncols) {  // the chunk's last columns
                    {
                        std::lock_guard<std::mutex> lk(s->rp_mu);
                        s->seq_replayed.fetch_add(1, std::memory_order_release);
                    }
                    s->rp_cv_done.notify_all();
                }
                if (s->rp_idle.load(std::memory_order_relaxed) > 0) s->rp_cv_work.notify_one();  // the group may have more
                continue;
            }
            // nothing to do: the GPU is behind (or other workers hold the groups that have work)
            if (!hungry) {
                hungry = true;
                s->rp_hungry.fetch_add(1, std::memory_order_relaxed);
            }
            // (the feed's final chunk count, stored before the flag: never a fresh read of seq_published)
            if (s->final_feed.load(std::memory_order_relaxed) && s->rp_all_published.load(std::memory_order_acquire) &&
                pop_ahead(s, s->n_groups.load(std::memory_order_acquire), s->rp_final_pub.load(std::memory_order_relaxed)))
                continue;
            if (++idle_spins < 64) {
                for (int i = 0; i < 32; i++) __builtin_ia32_pause();
                continue;
            }
            std::unique_lock<std::mutex> lk(s->rp_mu);
            if (s->rp_quit.load(std::memory_order_acquire)) {
                lk.unlock();
                drain();
                break;
            }
            s->rp_idle.fetch_add(1, std::memory_order_relaxed);
            s->rp_cv_work.wait_for(lk, std::chrono::microseconds(200));
            s->rp_idle.fetch_sub(1, std::memory_order_relaxed);
        }
    } catch (...) {
        s->rp_failed.store(true, std::memory_order_release);
    }
    if (hungry) s->rp_hungry.fetch_sub(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(s->rp_mu);
    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] worker %zu (cpu %d) done: busy %.2f ms, %llu units, %llu pushes\n", s->t_ms(), w, sched_getcpu(), (double)acc.busy_ns * 1e-6, (unsigned long long)acc.units, (unsigned long long)acc.pushes);
    s->rp_acc.pushes += acc.pushes;
    s->rp_acc.cands += acc.cands;
    s->rp_acc.busy_ns += acc.busy_ns;
    s->rp_acc.units += acc.units;
    s->rp_max_busy_ns = std::max(s->rp_max_busy_ns, acc.busy_ns);
    s->rp_min_busy_ns = std::min(s->rp_min_busy_ns, acc.busy_ns);
}

// The GPU runs up to n_slots chunks ahead of the host replay. Heap pushes are front-loaded (60 % of them
// belong to the first 10 % of a 100 M-row table) and inherently serial per column, so the host lags during
// that part; instead of idling, the GPU keeps scoring later chunks against the thresholds it has (staler
// thresholds only mean more records for the host to discard, never a wrong result) and the replay catches
// up while the long steady chunks run.
void feed_device_impl(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row) {
    const uint64_t stride = 1 + s->W_f;
    uint64_t pos = 0;
    s->t_feed0 = std::chrono::steady_clock::now();
    if (s->trace) fprintf(stderr, "[kgwas t=0.000] feed %llu rows\n", (unsigned long long)n_rows);
    if (s->count_patterns) hash_patterns(s, d_rows, n_rows);
    const uint64_t depth = s->direct ? (uint64_t)s->n_slots : 1;  // squeezed mode has a single squeeze buffer
    uint64_t sub = 0, cpy = 0, pub = 0;  // chunks submitted / record copies ordered / published in this feed
    // Chunks behind which the columns in select mode look at their pools for ties (~80 us per column): those that cross 40 and
    // 65 % of a large feed - a column that will need the exact replay (~5 ms for 10^5 pushes) gets it while its worker still has
    // slack beside the GPU; a tie found later than that would only move those 5 ms from kgwas_scan_finish, where the columns
    // with ties are replayed side by side, to the end of the feed (measured: checks at 87 / 94 / 97 % left the step where it
    // was and cost 40 ms of CPU) - and, for tables that arrive in many small feeds, those that take the session's row count
    // past the next power of 1.25.
    auto flag_tie_check = [&](Slot& sl, uint64_t pos_before, uint64_t c) {
        sl.tie_check = false;
        if (!s->lazy_any.load(std::memory_order_relaxed) || s->lazy_log_mode) return;
        // (whether a flagged chunk's columns ARE looked at is the worker's decision when it gets there: replay_group)
        if (n_rows >= (16ull << 20) && tie_checks_mode() != 0) {
            for (const double f : {0.4, 0.65}) {
                const uint64_t at = (uint64_t)(f * (double)n_rows);
                if (pos_before < at && pos_before + c >= at) sl.tie_check = true;
            }
        }
        if (s->rows_submitted + c >= s->tie_check_rows) {
            if (n_rows < (16ull << 20)) sl.tie_check = true;
            s->tie_check_rows = std::max<uint64_t>(4ull << 20, (uint64_t)(1.25 * (double)(s->rows_submitted + c)));
        }
    };
    bool running = false;
    std::chrono::steady_clock::time_point t_start;
    auto replayed = [&]() { return s->seq_replayed.load(std::memory_order_acquire); };
    auto start_async = [&]() {
        if (running) return;
        s->rp_quit.store(false, std::memory_order_release);
        s->rp_drain.store(false, std::memory_order_release);
        s->rp_acc = ReplayAcc();
        s->rp_max_busy_ns = 0;
        s->rp_min_busy_ns = ~0ull;
        t_start = std::chrono::steady_clock::now();
        s->pool->start(s->pool->size(), s->rp_fn);
        running = true;
    };
    auto wait_replayed = [&](uint64_t target) {
        std::unique_lock<std::mutex> lk(s->rp_mu);
        while (s->seq_replayed.load(std::memory_order_acquire) < target) {
            if (s->rp_failed.load(std::memory_order_acquire)) break;
            s->rp_cv_done.wait_for(lk, std::chrono::milliseconds(1));
        }
    };
    auto stop_async = [&]() {
        if (!running) return;
        {
            std::lock_guard<std::mutex> lk(s->rp_mu);
            // (every chunk of a final feed is replayed: what is left of the columns' result lists is made before the workers go)
            s->rp_drain.store(s->final_feed.load(std::memory_order_relaxed) && s->rp_all_published.load(std::memory_order_acquire) &&
                                  replayed() >= sub && !s->rp_failed.load(std::memory_order_acquire) && !exp_str("KGWAS_NO_DRAIN"),
                              std::memory_order_release);
            s->rp_quit.store(true, std::memory_order_release);
        }
        s->rp_cv_work.notify_all();
        s->pool->wait(false);
        running = false;
        add_replay_stats(s, s->rp_acc);
        // the replay's share of the wall clock: the busiest worker's time (they run side by side)
        s->st.replay_ms += (double)s->rp_max_busy_ns * 1e-6;
        if (s->rp_min_busy_ns != ~0ull) s->st.replay_min_ms += (double)s->rp_min_busy_ns * 1e-6;
        s->st.replay_wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        s->st.replay_splits += s->n_splits.exchange(0) + s->n_floated.exchange(0);
        s->st.columns_popped_ahead += s->n_popped_ahead.exchange(0);
        if (s->trace)
            fprintf(stderr, "[kgwas] replay ticks: scanning records %.1f M, heap updates %.1f M (TSC, all workers)\n",
                    (double)s->prof_scan.exchange(0) * 1e-6, (double)s->prof_heap.exchange(0) * 1e-6);
        if (s->trace)
            fprintf(stderr, "[kgwas] streaming replay: %llu units, cpu %.2f ms on %zu workers, busiest worker %.2f ms, wall %.2f ms\n",
                    (unsigned long long)s->rp_acc.units, (double)s->rp_acc.busy_ns * 1e-6, s->pool->size(),
                    (double)s->rp_max_busy_ns * 1e-6,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
        if (s->rp_failed.load(std::memory_order_acquire)) throw Error(KGWAS_ERR_NOMEM, "replay worker failed (out of memory?)");
    };
    // (sequence numbers restart with every feed: nothing is in flight between feeds)
    s->seq_submitted.store(0);
    s->seq_published.store(0);
    s->seq_replayed.store(0);
    if (s->ring_keep.load(std::memory_order_relaxed))
        s->ring_tail = 0;  // (the records of earlier feeds stay: the columns' logs refer to them)
    else
        s->ring_head = s->ring_tail = 0;
    s->ring_feed_start = s->ring_head;
    s->ring_freed = 0;
    // the session's own column groups (a feed may have split some, split_group)
    for (size_t g = 0; g < s->n_groups0 + (size_t)s->n_pheno; g++) {
        if (g < s->n_groups0)
            s->grp_cols[g] = s->grp_cols0[g];
        else
            s->grp_cols[g].clear();
        s->grp_owner[g].store(g < s->n_groups0 ? s->grp_home[g] : -1);
        s->gstate[g].done.store(0);
        s->gstate[g].busy.store(0);
    }
    s->n_groups.store(s->n_groups0);
    s->rp_hungry.store(0);
    s->rp_all_published.store(false);
    // rows are about to reach the heaps: result lists popped ahead of an earlier finish are history
    for (uint64_t j = 0; j < s->n_pheno; j++) s->col_popped[j].store(0, std::memory_order_relaxed);
    s->final_feed.store(s->final_feed_next, std::memory_order_release);
    s->final_feed_next = false;
    try {
        for (;;) {
            if (pos < n_rows && !s->all_full) {  // dense phase: until every heap is full
                wait_replayed(sub);  // (nothing is in flight here in practice: heaps never un-fill)
                stop_async();
                const uint64_t c = std::min<uint64_t>(s->dense_chunk, n_rows - pos);
                // First dense chunk of an empty session with sparse work behind it: the device also selects each
                // column's topn-th largest score of the chunk - the minimum its heap WILL have once these rows are
                // pushed - and takes it as its threshold, so the first sparse chunks are submitted before the host
                // pushes the 1.2 M (2.4 M at 201 columns) dense scores: the GPU filters while the heaps fill (2 ms of
                // a 28 ms step at 101 columns, 8 of 126 at 250 M rows x 201).
                static const bool no_overlap = exp_set("KGWAS_NO_DENSE_OVERLAP");  // experiments
                bool empty = s->coarse && !no_overlap && n_rows - pos > c && sub == 0;
                for (uint64_t j = 0; j < s->n_pheno && empty; j++) empty = s->heaps[j].size() == 0 && s->lazy[j].n_logged == 0;
                if (empty) {
                    const auto td0 = std::chrono::steady_clock::now();
                    run_dense(s, d_rows + pos * stride, c, first_row + pos, nullptr, nullptr, true, /*select=*/true);
                    const uint64_t dense_first = first_row + pos;
                    pos += c;
                    // every heap will be full after these rows, and no score is a NaN (the heap's order with NaNs in it is
                    // not a total one: plain path)
                    const bool early = s->h_sel_info.p[0] >= s->max_topn && s->h_sel_info.p[1] == 0;
                    const auto submit_one = [&]() {
                            const uint64_t cs = std::min<uint64_t>(next_sparse_chunk(s), n_rows - pos);
                            const size_t si = (size_t)(sub % (uint64_t)s->n_slots);
                            s->slot_left[si].store((uint32_t)s->n_pheno, std::memory_order_release);  // (columns: groups may be split)
                            s->slack_rows = n_rows - pos - cs;
                            flag_tie_check(s->slot[si], pos, cs);
                            submit_sparse(s, s->slot[si], d_rows + pos * stride, cs, first_row + pos, /*count_hist=*/true);
                            s->slack_rows = 0;
                            s->rows_submitted += cs;
                            sub++;
                            s->seq_submitted.store(sub, std::memory_order_release);
                            pos += cs;
                            if (s->trace) fprintf(stderr, "[kgwas t=%.3f] presubmitted chunk %llu (%llu rows), fill %s\n", s->t_ms(), (unsigned long long)(sub - 1), (unsigned long long)cs, s->pool->finished() ? "done" : "running");
                    };
                    if (early && pos < n_rows) {
                        // the first sparse chunk goes out before the dense scores are in host memory (their copy runs beside it)
                        s->sel_valid = true;
                        s->rows_submitted = std::max(s->rows_submitted, s->rows_done + c);
                        submit_one();
                        s->sel_valid = false;
                    }
                    wait_dense_copy(s);
                    const std::function<void()> presubmit = [&]() {  // (the replay workers start after the fill)
                        s->sel_valid = true;
                        s->rows_submitted = std::max(s->rows_submitted, s->rows_done + c);
                        // GPU work for the duration of the fill: at least two chunks, more only while the workers are
                        // still pushing (a submission costs this thread 0.1-0.2 ms; the replay starts when both are done)
                        while (pos < n_rows && sub < std::min<uint64_t>(depth, 12) && (sub < 2 || !s->pool->finished())) submit_one();
                        s->sel_valid = false;
                        // While the workers go on pushing: order the record copies of the chunks whose counts have come in,
                        // so the first chunks' records are in host memory when the fill ends (the copy of chunk 0 - 28 MB
                        // at 101 columns - used to start only then: 0.7 ms in which sixteen workers had nothing to replay).
                        while (!s->pool->finished() && cpy < sub) {
                            Slot& sl = s->slot[(size_t)(cpy % (uint64_t)s->n_slots)];
                            if (sl.used_coarse && hipEventQuery(sl.ev_counts) != hipSuccess) {
                                // (sleeping, not spinning: this thread is not pinned and may share a CPU with a pinned worker)
                                std::this_thread::sleep_for(std::chrono::microseconds(20));
                                continue;
                            }
                            if (!fetch_records(s, sl, cpy)) break;  // the record ring is full: the main loop deals with it
                            if (s->trace) fprintf(stderr, "[kgwas t=%.3f] counts of chunk %llu in, record copy ordered (during the fill)\n", s->t_ms(), (unsigned long long)cpy);
                            cpy++;
                        }
                    };
                    dense_fill(s, c, dense_first, td0, early ? &presubmit : nullptr);
                    if (sub) start_async();  // chunks are in flight: the replay workers take over from the fill
                    continue;
                }
                run_dense(s, d_rows + pos * stride, c, first_row + pos, nullptr, nullptr, true);
                pos += c;
                continue;
            }
            // Submit ONE chunk per turn of this loop while slots are free, then look (without blocking) for counts that
            // have arrived and copies that have landed: a submission costs this thread 0.1-0.2 ms, and submitting every
            // free slot's chunk first - 24 of them at 100 M rows - kept the first chunk's records from the replay for
            // 2-3 ms after the GPU had delivered them. The blocking waits below are only taken when nothing can be
            // submitted.
            // ONE read of the replay's progress per turn: the decisions below and the waits' targets must come from the
            // same value (a target taken from a fresher read can lie beyond everything that is published - the wait
            // would then never end)
            const uint64_t rep_now = replayed();
            if (pos < n_rows && sub - rep_now < depth) {
                start_async();
                const uint64_t c = std::min<uint64_t>(next_sparse_chunk(s), n_rows - pos);
                const size_t si = (size_t)(sub % (uint64_t)s->n_slots);  // its previous chunk was replayed n_slots chunks ago
                s->slot_left[si].store((uint32_t)s->n_pheno, std::memory_order_release);  // (columns: groups may be split)
                s->slack_rows = n_rows - pos - c;
                flag_tie_check(s->slot[si], pos, c);
                submit_sparse(s, s->slot[si], d_rows + pos * stride, c, first_row + pos, /*count_hist=*/true);
                s->slack_rows = 0;
                if (s->trace) fprintf(stderr, "[kgwas t=%.3f] submit chunk %llu (%llu rows)\n", s->t_ms(), (unsigned long long)sub, (unsigned long long)c);
                s->rows_submitted += c;
                sub++;
                s->seq_submitted.store(sub, std::memory_order_release);
                pos += c;
            }
            // Coarse chunks hand their records over in two steps: counts first (compute stream), then exactly that many
            // records on the copy stream, ordered here as soon as the counts are in - before waiting for an older
            // chunk's copy if this chunk's counts are already there, so the copy engine never waits for this thread.
            const bool can_submit = pos < n_rows && sub - rep_now < depth;
            // (exact-scorer chunks have no counts step and no ev_counts: querying a null event fails AND leaves the error
            // for the next hipGetLastError - the next kernel launch then reports "invalid resource handle")
            if (cpy < sub && ((cpy == pub && !can_submit) || !s->slot[(size_t)(cpy % (uint64_t)s->n_slots)].used_coarse ||
                              hipEventQuery(s->slot[(size_t)(cpy % (uint64_t)s->n_slots)].ev_counts) == hipSuccess)) {
                if (fetch_records(s, s->slot[(size_t)(cpy % (uint64_t)s->n_slots)], cpy)) {
                    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] counts of chunk %llu in, record copy ordered\n", s->t_ms(), (unsigned long long)cpy);
                    cpy++;
                    continue;
                }
                // the record ring is full: publish what is fetched; if all of that is published, wait for the replay
                if (pub == cpy && s->ring_keep.load(std::memory_order_relaxed)) {
                    wait_replayed(pub);
                    if (s->rp_failed.load(std::memory_order_acquire)) break;
                    ring_to_recycling(s, pub);
                    continue;
                }
                if (pub == cpy) {
                    wait_replayed(rep_now + 1);
                    if (s->rp_failed.load(std::memory_order_acquire)) break;
                    continue;
                }
            }
            if (pub < cpy && (!can_submit || hipEventQuery(s->slot[(size_t)(pub % (uint64_t)s->n_slots)].ev_done) == hipSuccess)) {
                // publish the oldest chunk the GPU still owes
                Slot& sl = s->slot[(size_t)(pub % (uint64_t)s->n_slots)];
                wait_event(s, sl.ev_done);  // records and counts are in host memory
                if (chunk_complete(s, sl)) {
                    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] publish chunk %llu\n", s->t_ms(), (unsigned long long)pub);
                    pub++;
                    {
                        std::lock_guard<std::mutex> lk(s->rp_mu);
                        s->seq_published.store(pub, std::memory_order_release);
                    }
                    // The feed's last chunk. Only AFTER seq_published holds the final count: a worker that sees the flag
                    // (acquire) must not pair it with the count before this chunk - every caught-up group would look
                    // complete one chunk early and pop_ahead would pop heaps this chunk's records still change.
                    if (pos >= n_rows && pub == sub) {
                        s->rp_final_pub.store(pub, std::memory_order_relaxed);
                        s->rp_all_published.store(true, std::memory_order_release);
                    }
                    s->rp_cv_work.notify_all();
                    // The exact minima as far as the workers have come (racy reads of monotone values: any value a
                    // minimum has had after some prefix of the rows is a valid bound for every later row).
                    upload_thresholds(s);
                } else {
                    // A list overflowed. Everything before the chunk is replayed first, the younger chunks finish on
                    // the GPU (their records stay in their slots and are published in order afterwards), and this
                    // range is redone synchronously in halves against the heaps' exact minima.
                    wait_replayed(pub);
                    stop_async();
                    KGWAS_HIP(hipStreamSynchronize(s->stream));
                    process_range_sync(s, sl, sl.rows, sl.n_rows, sl.first_row);
                    pub++;
                    for (size_t g = 0; g < s->n_groups.load(); g++) s->gstate[g].done.store(pub, std::memory_order_release);
                    s->seq_replayed.store(pub, std::memory_order_release);
                    s->seq_published.store(pub, std::memory_order_release);
                    if (pub < sub || pos < n_rows) start_async();
                }
                continue;
            }
            if (can_submit) continue;
            if (pos < n_rows) {  // every slot holds a chunk that is still being replayed
                wait_replayed(rep_now + 1);
                if (s->rp_failed.load(std::memory_order_acquire)) break;
                continue;
            }
            break;
        }
        {
            auto w0 = std::chrono::steady_clock::now();
            wait_replayed(sub);
            s->st.replay_tail_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        }
        stop_async();
    } catch (...) {
        if (running) {
            {
                std::lock_guard<std::mutex> lk(s->rp_mu);
                s->rp_quit.store(true, std::memory_order_release);
            }
            s->rp_cv_work.notify_all();
            s->pool->wait(false);
        }
        (void)hipStreamSynchronize(s->stream);
        throw;
    }
    s->rows_done = s->rows_submitted;
    s->st.rows_fed += n_rows;
}

}  // namespace kgwas
