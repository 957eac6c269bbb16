/* include/kgwas.h — C ABI of libkgwas: the MI355X-native k-mer association-scan engine.
 *
 * Drop-in boundary for ONE hot path of voichek/kmersGWAS: the associate_kmers /
 * emma_kinship_kmers scan. The reference has no plugin or FFI interface — its boundary is
 * the process (argv + files) and, inside the binary, the method surface of
 * MultipleKmersDataBases and BestAssociationsHeap. Each entry point below names the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions: plain C, no exceptions cross the boundary. Every function returns
 * KGWAS_OK (0) or a negative KGWAS_ERR_*; kgwas_last_error() gives the thread-local
 * message. Handles are created and freed by the library. Caller-owned input buffers are
 * only read during the call. Result pointers stay valid until the owning handle is
 * destroyed. All compute runs on the GPU through hand-written gfx950 kernels: there is
 * NO CPU fallback — without a usable HIP device the compute entry points fail with
 * KGWAS_ERR_DEVICE.
 */
#ifndef KGWAS_H
#define KGWAS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGWAS_OK 0
#define KGWAS_ERR_ARG (-1)    /* bad argument */
#define KGWAS_ERR_IO (-2)     /* file cannot be opened / read / written */
#define KGWAS_ERR_FORMAT (-3) /* a reference std::logic_error: bad magic, k, size, unknown accession ... */
#define KGWAS_ERR_DEVICE (-4) /* no HIP device or a HIP call failed */
#define KGWAS_ERR_STATE (-5)  /* call out of order */
#define KGWAS_ERR_NOMEM (-6)

const char* kgwas_last_error(void);
int kgwas_version(void);
int kgwas_device_count(int* n_devices);
/* CPUs this process may really keep busy: the cgroup CPU quota if there is one, else the hardware thread count. What
 * kgwas_scan_params.host_threads = 0 resolves to; the command-line tools raise a smaller --parallel to it (the
 * reference's --parallel sizes a pool of scoring tasks, kmers_gwas.py passes 1 by default - pipeline_parser.py:31 -
 * and here the threads replay heap pushes beside the GPU: one thread would be ~20x slower than the GPU it serves). */
uint32_t kgwas_host_cpu_quota(void);

/* ------------------------------------------------------------------------------------
 * .table / .names reader.
 * Replaces MultipleKmersDataBases::MultipleKmersDataBases header guards
 * (src/kmers_multiple_databases.cpp:39-94), load_kmers_talbe_column_names
 * (src/kmer_general.cpp:45-53) and create_map_from_all_DBs (:297-311).
 * kmer_len == 0 skips the k check (for tools that do not know k).
 * ---------------------------------------------------------------------------------- */
typedef struct kgwas_table kgwas_table;
int kgwas_table_open(const char* base, uint32_t kmer_len, kgwas_table** out);
int kgwas_table_info(const kgwas_table* t, uint64_t* n_acc_file, uint64_t* n_rows, uint64_t* words_per_row,
                     uint32_t* kmer_len);
int kgwas_table_name(const kgwas_table* t, uint64_t i, const char** name);
/* col_out[i] = file column of accession acc[i]; unknown or duplicated name -> KGWAS_ERR_FORMAT
 * (intersect_phenotypes_to_present_DBs with must_be_present, src/kmer_general.cpp:239-253;
 *  get_index_DB :227-237). */
int kgwas_table_column_map(const kgwas_table* t, const char* const* acc, uint64_t n, uint64_t* col_out);
/* Raw rows [u64 kmer][W_f u64] of file rows [row0, row0+n) into dst (host). */
int kgwas_table_read_rows(kgwas_table* t, uint64_t row0, uint64_t n, uint64_t* dst);
void kgwas_table_close(kgwas_table* t);

/* ------------------------------------------------------------------------------------
 * Phenotype TSV. Replaces load_phenotypes_file (src/kmer_general.cpp:175-205).
 * ---------------------------------------------------------------------------------- */
typedef struct kgwas_pheno kgwas_pheno;
int kgwas_pheno_load(const char* path, kgwas_pheno** out);
int kgwas_pheno_info(const kgwas_pheno* p, uint64_t* n_pheno, uint64_t* n_acc);
int kgwas_pheno_name(const kgwas_pheno* p, uint64_t j, const char** name);
int kgwas_pheno_accession(const kgwas_pheno* p, uint64_t i, const char** acc);
/* Y: n_pheno x n_acc float32, row-major, accession (file) order of the TSV. */
int kgwas_pheno_values(const kgwas_pheno* p, const float** Y);
void kgwas_pheno_free(kgwas_pheno* p);

/* min_count = max(ceil(n_acc*maf), mac) (src/associate_kmers.cpp:99-103). */
uint64_t kgwas_min_count(uint64_t n_acc, double maf, uint64_t mac);

/* ------------------------------------------------------------------------------------
 * BestAssociationsHeap (src/best_associations_heap.h:32-54, .cpp:26-127): bounded min-heap
 * with the reference's strict-'>' replacement. NOT a std::priority_queue: csrc/heap.h performs,
 * by hand, the element moves libstdc++'s std::push_heap / std::pop_heap (__push_heap, __adjust_heap)
 * make for the reference's comparator (a.score > b.score, src/kmer_general.h:113-128) - on 16-byte
 * (score, slot) entries, several heaps in lockstep, integer compares where all scores are >= +0 -
 * so that ties resolve identically: which equal-score entry survives at the boundary and the order
 * equal scores pop in. The emulation is pinned against a literal std::priority_queue over the
 * reference's tuple type (oracle/oracle.cpp) on tie-heavy, NaN, negative and +inf streams, in the
 * CPU suite (tests/test_host.py::test_heap_mirror_equals_oracle_heap_under_ties) and again on the
 * GPU box (tests/test_gpu_parity.py::test_heap_mirror_equals_std_priority_queue_on_the_gpu_box);
 * checked against libstdc++ of GCC 11.4 (GLIBCXX_3.4.30), the toolchain of this image and of the
 * test boxes. A libstdc++ whose heap algorithms move elements differently would fail those tests;
 * the reference itself would then produce different tie orders with it.
 * Used by the scan for its host-side replay and exposed for cross-shard merges.
 * RUN-TIME GUARD (csrc/heap_guard.cpp): once per process, before the first heap or session exists, a fixed stream of ties, NaNs,
 * negative and infinite scores goes through the emulation (single pushes and the lockstep form) and through a literal
 * std::priority_queue over the reference's tuple / comparator of THE LIBSTDC++ THIS PROCESS RUNS WITH; on any difference
 * kgwas_heap_new, kgwas_scan_create, kgwas_multiscan_create and kgwas_snps_* return KGWAS_ERR_STATE. kgwas_heap_selfcheck(0) runs
 * (or re-reports) that check; kgwas_heap_selfcheck(1) is a test hook that holds the emulation against a reference with a
 * DIFFERENT tie rule and must therefore fail.
 * ---------------------------------------------------------------------------------- */
int kgwas_heap_selfcheck(uint32_t flags);
typedef struct kgwas_heap kgwas_heap;
int kgwas_heap_new(uint64_t max_results, kgwas_heap** out);
/* add_association for n entries in the given order. */
int kgwas_heap_add_many(kgwas_heap* h, const uint64_t* kmer, const double* score, const uint64_t* row, uint64_t n);
int kgwas_heap_size(const kgwas_heap* h, uint64_t* size, uint64_t* insertions, double* lowest);
/* output_to_file_with_scores order: ascending pops from a copy; arrays hold `size` entries. */
int kgwas_heap_pop_all(const kgwas_heap* h, uint64_t* kmer, double* score, uint64_t* row);
/* get_kmers_for_output: (kmer, rank, row) sorted by row; rank = queue size at pop (best = 1). */
int kgwas_heap_output_list(const kgwas_heap* h, uint64_t* kmer, uint64_t* rank, uint64_t* row);
/* get_rows_sorted_indices (src/best_associations_heap.cpp:135-147): the entries' row indices, ascending; `size` entries. */
int kgwas_heap_rows_sorted(const kgwas_heap* h, uint64_t* row);
/* output_to_file (:65-74, with_scores = 0: the k-mers as 8 bytes each) / output_to_file_with_scores (:80-90, with_scores = 1:
 * k-mer + score, 16 bytes each), in ascending pop order; the file is created or truncated. */
int kgwas_heap_output_to_file(const kgwas_heap* h, const char* path, int with_scores);
void kgwas_heap_free(kgwas_heap* h);

/* Test hook: one phenotype column's select-or-replay decision on the host alone (scan_lazy.cpp). The records
 * of n_chunks chunks (chunk c: chunk_n[c] records, rows chunk_row0[c] + row_in_chunk[i], ascending; a score of -inf is a
 * placeholder and no record) are logged - by reference into the caller's arrays (by_ref != 0) or by copy, the logs detached from
 * the caller's arrays behind chunk `detach_after_chunk` (< 0: never) - with chunk_thr[c] as the device's threshold behind chunk c
 * (any lower bound of the N-th largest score up to there; 0: none). Then the column is finished: *selected = 1 if its N + 1
 * largest scores were pairwise distinct, none NaN or negative, and the lists were made by selection; 0 if its log was replayed
 * through the heap mirror. out_*: the result lists in ascending pop order (room for topn entries), *out_n their length. */
int kgwas_select_check(uint64_t topn, uint64_t n_chunks, const uint64_t* chunk_n, const uint64_t* chunk_row0, const double* chunk_thr,
                       const double* score, const uint64_t* kmer, const uint32_t* row_in_chunk, int by_ref, int64_t detach_after_chunk,
                       int* selected, uint64_t* out_kmer, double* out_score, uint64_t* out_row, uint64_t* out_n);

/* ------------------------------------------------------------------------------------
 * Association scan session = pass 1 of associate_kmers (src/associate_kmers.cpp:99-148):
 * MultipleKmersDataBases::load_kmers (MAC filter + squeeze, :103-146),
 * add_kmers_to_heap / calculate_kmer_score (:275-284, :327-363) for every phenotype column,
 * feeding one BestAssociationsHeap per column. Rows are fed in file order, in any number of
 * feed calls; results do not depend on how the rows are split.
 * ---------------------------------------------------------------------------------- */
typedef struct kgwas_scan kgwas_scan;

#define KGWAS_KERNEL_AUTO 0
#define KGWAS_KERNEL_VALU 1 /* exact-order select+add on the vector ALU */
#define KGWAS_KERNEL_MFMA 2 /* exact-order f32 MFMA (v_mfma_f32_16x16x4_f32) */
#define KGWAS_KERNEL_COARSE 3 /* matrix-pipe coarse filter (block-scaled FP4 x FP6/FP4, or int8 with KGWAS_COARSE_MX=0) with a rigorous bound + exact re-scoring of survivors;
                                the dense phase and overflow re-runs use an exact kernel. Same results. */
#define KGWAS_KERNEL_NARROW 4 /* reported in kgwas_scan_stats.kernel_used only: the filter of scans with 1-4 phenotype
                                columns (FP4 table bits x FP8 phenotype slices on the block-scaled MFMA; requested
                                through KGWAS_KERNEL_AUTO or _COARSE), exact re-scoring as above. Same results. */

typedef struct kgwas_scan_params {
    uint32_t struct_size;    /* sizeof(kgwas_scan_params) */
    int32_t device;          /* HIP device ordinal */
    uint64_t n_acc_file;     /* S_f: accessions (bit columns) in the table */
    uint64_t n_acc;          /* S: phenotyped accessions */
    const uint64_t* col;     /* [S] file column of phenotyped accession i (phenotype order) */
    uint64_t n_pheno;        /* phenotype columns (1 + permutations) */
    const float* Y;          /* n_pheno x S float32 row-major, phenotype order */
    const uint64_t* topn;    /* [n_pheno] heap sizes (-n / --first_phenotype_best) */
    uint64_t min_count;      /* effective minor allele count */
    uint64_t chunk_rows;     /* max rows per device chunk; 0 = default */
    uint32_t host_threads;   /* replay threads; 0 = hardware concurrency */
    uint32_t kernel;         /* KGWAS_KERNEL_* */
    uint32_t record_history; /* what a later shard must keep for a cross-shard merge: 1 = every effective heap push
                                (kgwas_scan_history, kgwas_scan_history_above); 2 = only each heap's last evictions
                                (kgwas_scan_history_above alone; it fails loudly if the ring was too short) */
    uint32_t count_patterns; /* --pattern_counter: count distinct presence/absence patterns of tested rows */
} kgwas_scan_params;

typedef struct kgwas_scan_stats {
    uint64_t rows_fed;          /* file rows seen */
    uint64_t rows_tested;       /* rows passing the MAC filter (= .tested_kmers) */
    uint64_t candidates;        /* (row, phenotype) records shipped device -> host */
    uint64_t heap_pushes;       /* effective pushes over all heaps */
    uint64_t chunks;            /* device chunks launched */
    uint64_t score_launches;    /* launches of the scoring kernel */
    double score_kernel_ms;     /* sum of hipEvent durations of the scoring kernel */
    double squeeze_kernel_ms;   /* sum of hipEvent durations of the squeeze kernel (0 in direct mode) */
    double replay_ms;           /* host time spent replaying candidates: the busiest replay worker's busy time (the workers
                                   run beside the GPU and beside each other) + the synchronous dense / overflow replays */
    double gpu_wait_ms;         /* host wall time blocked waiting for sparse chunks to finish */
    double dense_ms;            /* host wall time of the dense (heap-filling) phase, GPU + replay */
    double coarse_kernel_ms;    /* sum of hipEvent durations of coarse_kernel alone (KGWAS_KERNEL_COARSE) */
    uint64_t coarse_launches;   /* its launches */
    uint32_t kernel_used;       /* KGWAS_KERNEL_VALU, _MFMA, _COARSE or _NARROW */
    uint32_t direct_mode;       /* 1 = scorer read the file layout in place (no squeeze pass) */
    uint64_t patterns;          /* distinct pattern hashes among tested rows (count_patterns; valid after finish) */
    /* coarse filter, per operand set: [0] = one int8 slice per phenotype column, [1] = two slices */
    uint32_t coarse_mode_tiles[2];     /* 16-column int8 operand tiles per LDS group of the set's main launch (0 = set not built) */
    uint32_t coarse_mode_lgroups[2];   /* LDS groups a row passes through (all launches of the set) */
    uint64_t coarse_mode_launches[2];
    uint64_t coarse_mode_rows[2];      /* rows filtered with this set */
    double coarse_mode_ms[2];          /* hipEvent time of its coarse_kernel launches */
    double replay_cpu_ms;       /* CPU time of the replay summed over the workers (replay_ms: the busiest worker's share) */
    double replay_tail_ms;      /* wall time the replay still needed after the GPU had finished the feed's last chunk */
    uint32_t coarse_mode_tile_slices[2]; /* operand tiles a row is multiplied with, over all LDS groups and launches of the set */
    uint32_t coarse_mx;         /* 1 = the filter is the block-scaled one (score_mx.hip: FP4 table bits x FP6 / FP4 slices on
                                   v_mfma_scale_f32_16x16x128_f8f6f4; coarse_mode_tiles then counts COLUMN tiles, each carrying
                                   all slices of its 16 columns), 0 = the int8 one (KGWAS_COARSE_MX=0) */
    uint32_t coarse_mx_s1_fp6;  /* block-scaled filter: the second slice is FP6 (else FP4) */
    uint32_t coarse_mx_steps;   /* block-scaled filter: MFMA steps (K = 128) per row tile, column tile and slice */
    uint32_t replay_threads;    /* host threads replaying heap pushes in this session */
    double replay_min_ms;       /* the LEAST busy replay worker's busy time (replay_ms: the busiest one's; replay_cpu_ms / replay_threads: the mean) */
    double replay_wall_ms;      /* wall time from the start of the streaming replay (first sparse chunk submitted) to its end */
    uint64_t replay_splits;     /* column groups of the replay that left their worker because they had fallen behind (a slow or
                                   shared CPU under it): handed whole to the workers that are ahead, or cut into single
                                   columns once other workers had nothing left to do */
    uint64_t columns_popped_ahead; /* columns whose result lists were made by idle replay workers at the end of the last feed
                                      (kgwas_scan_expect_finish) instead of by kgwas_scan_finish */
    uint32_t coarse_mx32;       /* always 0 since round 6: the 32 x 32 x 64 form of the block-scaled filter (built, verified and measured
                                   slower in round 4: docs/HISTORY.md) was removed; the field keeps the struct's layout */
    uint32_t coarse_mx_stream;  /* block-scaled filter in its operand-streaming form (score_mxs.hip: all column tiles of an operand
                                   group per wave, operands through an LDS ring; every row loaded and expanded once per group):
                                   1 = the default shapes (one column group of up to 7 tiles, or two of them side by side in a
                                   block), 2 / 3 = one column group of up to 13 tiles with eight waves of 32 rows / four waves of 64
                                   (KGWAS_MXS_FORM=1 / 2). This field replaces the former `reserved0`: the struct's size and
                                   layout are unchanged. */
    /* -- added in ABI version 5 (kgwas_abi_version): the struct grew by 8 bytes -- */
    uint32_t columns_selected;  /* columns whose result lists kgwas_scan_finish made by SELECTION: their N largest scores (and the
                                   N + 1-th) were pairwise distinct, none NaN or negative, so the lists are the N largest in ascending
                                   order whatever libstdc++'s heap layout; such columns' candidates were logged, not replayed, and
                                   heap_pushes does not count them (KGWAS_FULL_REPLAY=1: every column is replayed, as before) */
    uint32_t columns_replayed_at_finish; /* columns that were in select mode until finish and then needed the exact replay of their
                                   log (a tie among their N + 1 largest scores that did not show in the first dense chunk) */
} kgwas_scan_stats;

/* Version of this header's struct layouts and entry points. A caller built against an older header must not pass its (smaller)
 * kgwas_scan_stats to a newer library: compare KGWAS_ABI_VERSION with kgwas_abi_version() at start-up.
 * Version 6 (round 6): no struct changed; new entry points kgwas_heap_selfcheck, kgwas_scan_select_mode and the test hook
 * kgwas_scan_debug_residuals; kgwas_scan_lowest takes a non-const session (it always mutated it); kgwas_scan_stats.coarse_mx32 is
 * always 0 (the 32 x 32 x 64 filter form was removed). */
#define KGWAS_ABI_VERSION 6
uint32_t kgwas_abi_version(void);

int kgwas_scan_create(const kgwas_scan_params* p, kgwas_scan** out);
/* A hint, never needed for correctness: the NEXT feed is the last one before kgwas_scan_finish (the tools know it: they
 * feed a whole table). kgwas_scan_finish pops every heap into its result lists (output_to_file_with_scores' order,
 * src/best_associations_heap.cpp:82-92) - 10 001 pops per column; with the hint, replay workers that run out of records
 * near the end of that feed pop the columns that are complete while the slowest worker is still replaying, and finish
 * only does what is left. Anything that changes a heap afterwards (another feed, kgwas_scan_absorb,
 * kgwas_scan_heaps_import) discards those lists. */
int kgwas_scan_expect_finish(kgwas_scan* s);
/* Rows already resident in HBM (file layout, 8-byte aligned). hip_stream may be NULL. */
int kgwas_scan_feed_device(kgwas_scan* s, const void* d_rows, uint64_t n_rows, uint64_t first_row, void* hip_stream);
/* Rows in host memory (file layout). Chunked, double-buffered: a producer thread stages 128 MiB pieces into pinned
 * memory and piece k+1 is copied (own stream) while piece k is scored and replayed. */
int kgwas_scan_feed_host(kgwas_scan* s, const uint64_t* rows, uint64_t n_rows, uint64_t first_row);
/* Rows [row0, row0 + n_rows) straight from an open .table: the same pipeline with the producer thread reading the
 * file (pread) into the pinned pieces, so disk, PCIe and GPU + replay overlap. Replaces the reference's
 * load-a-batch-then-compute loop (src/associate_kmers.cpp:104-148); first_row of the feed is row0. */
int kgwas_scan_feed_table(kgwas_scan* s, kgwas_table* t, uint64_t row0, uint64_t n_rows);
int kgwas_scan_finish(kgwas_scan* s);
/* Heap of phenotype j in heap-pop (ascending score) order: rank of entry i is n - i.
 * row = file row index. Valid after kgwas_scan_finish. */
int kgwas_scan_result(kgwas_scan* s, uint64_t j, uint64_t* n, const uint64_t** kmer, const double** score,
                      const uint64_t** row);
/* Effective pushes of phenotype j in row order (needs record_history). */
int kgwas_scan_history(kgwas_scan* s, uint64_t j, uint64_t* n, const uint64_t** kmer, const double** score,
                       const uint64_t** row);
int kgwas_scan_get_stats(const kgwas_scan* s, kgwas_scan_stats* st);
/* Empty the heaps, histories and statistics but keep every device / pinned buffer, so a session can
 * scan another table (or the same one again) without paying allocation again. */
int kgwas_scan_reset(kgwas_scan* s);
void kgwas_scan_destroy(kgwas_scan* s);

/* Cross-shard merge without re-playing shard 0: current heap minima (lowest_score) and fullness of
 * every column, and in-order replay of later shards' (pre-filtered) histories INTO this scan's heaps.
 * Exactness: the heap of shard 0 is what a single scan holds after those rows; an entry of shard g >= 1
 * whose score is not above max(final minima of the full heaps of shards < g) is rejected by
 * add_association whenever it arrives, so the sender may drop it. counts/kmer/score/row as in
 * kgwas_merge_shards (shards in row order, all after this scan's rows). Call kgwas_scan_finish again.
 * kgwas_scan_lowest MUTATES the session (it is not a read-only query): for columns in select mode it brings their pools up to
 * date and may run a selection, and a column that saw a NaN or negative score is given its heap (its log replayed, heap_pushes
 * grows). It must not overlap a feed or another call on the same session.
 * kgwas_scan_select_mode: *on = 1 if the session keeps its columns in select mode (logs + pools, result lists by selection where
 * the scores decide) - i.e. a filter session without a full push log and without KGWAS_FULL_REPLAY=1 -, 0 if
 * every column is replayed as it streams (exact-scorer sessions among them). Merge layers use it to pick their route. */
int kgwas_scan_lowest(kgwas_scan* s, double* lowest, uint8_t* full);
int kgwas_scan_select_mode(const kgwas_scan* s, int* on);
/* Test hook (sessions created under KGWAS_DEBUG_RESIDUALS=1, else KGWAS_ERR_STATE): the quantisation residuals of column
 * `column` in phenotype-file order - resid_i = y_i - c - (the value the filter form's slices encode for sample i) - for form 0
 * (one-slice operand set), 1 (two-slice set: block-scaled FP6 + FP4 / FP6 + FP6 or int8, whichever the session built) or 2
 * (narrow filter, three FP8 slices). A row whose set bits are exactly the samples with resid_i > 0 attains
 * sum_i g_i resid_i = Rall: the filters' bound |yigi_ref - yc| <= Eg + min(Rall, N1 rmax) is tight there, which is where a test
 * has to look for lost pushes (tests/test_gpu_parity.py::test_adversarial_rows_at_the_filters_bound). */
int kgwas_scan_debug_residuals(const kgwas_scan* s, uint32_t form, uint64_t column, double* out);
int kgwas_scan_absorb(kgwas_scan* s, uint64_t n_shards, const uint64_t* counts, const uint64_t* const* kmer,
                      const double* const* score, const uint64_t* const* row);
/* The part of the recorded history that can still matter after heaps whose minima are thr[j]: entries with
 * score > thr[j] (thr[j] = -inf keeps all), column by column, flat: counts[j] entries of column j follow those of
 * column j-1. The arrays stay valid until the next call of this function or of kgwas_scan_heaps_export. */
int kgwas_scan_history_above(kgwas_scan* s, const double* thr, uint64_t* counts, const uint64_t** kmer,
                             const double** score, const uint64_t** row);
/* Column-distributed merge: the state of heaps cols[0..n_cols) in heap-array order (sizes[c] entries each, flat),
 * and its exact re-creation in another scan session of the same shape - layout included, so the heap goes on
 * there exactly as it would have here (a block of columns of a multi-GPU job is merged on one rank). */
int kgwas_scan_heaps_export(kgwas_scan* s, uint64_t n_cols, const uint64_t* cols, uint64_t* sizes, const uint64_t** kmer,
                            const double** score, const uint64_t** row);
int kgwas_scan_heaps_import(kgwas_scan* s, uint64_t n_cols, const uint64_t* cols, const uint64_t* sizes, const uint64_t* kmer,
                            const double* score, const uint64_t* row);
/* The same two exports as MESSAGES written straight into caller memory (the pinned staging buffer of an all-to-all):
 * message m covers the columns msg_col0[m] .. msg_col0[m] + msg_ncols[m] - 1 and is laid out as 64-bit words
 *   [words that follow][counts or sizes: msg_ncols[m]][kmer: T][score bit patterns: T][row: T],  T = sum of the counts
 * (msg_ncols[m] = 0: the one word 0). Messages follow each other in `out`; msg_words[m] receives message m's length.
 * Nothing is written unless the messages fit cap_words (the caller grows the buffer and calls again). */
int kgwas_scan_history_above_msgs(kgwas_scan* s, const double* thr, uint64_t n_msgs, const uint64_t* msg_col0,
                                  const uint64_t* msg_ncols, uint64_t* out, uint64_t cap_words, uint64_t* msg_words);
int kgwas_scan_heaps_export_msgs(kgwas_scan* s, uint64_t n_msgs, const uint64_t* msg_col0, const uint64_t* msg_ncols,
                                 uint64_t* out, uint64_t cap_words, uint64_t* msg_words);

/* ------------------------------------------------------------------------------------
 * The same scan over several GPUs of one node, inside one process (the reference's caller, kmers_gwas.py:133-148, runs
 * ONE associate_kmers binary): the rows are cut into n_devices contiguous shards in file order, shard g is scanned on
 * devices[g] by its own session and host thread (a device may be listed more than once), no data moves between the
 * devices during the scan, and the later shards' effective heap pushes that can still matter are then replayed, in
 * row order, into shard 0's heaps (kgwas_scan_history_above / kgwas_scan_absorb). Results are those of a single scan.
 * p->device is ignored; p->host_threads is the replay-thread budget of the whole job (divided among the shards);
 * p->record_history applies to shard 0 (later shards keep what the merge needs; a shard whose eviction ring turns out
 * too short is scanned again with the full log). kgwas_multiscan_run_* may be called repeatedly with consecutive row
 * ranges (not with count_patterns); call kgwas_multiscan_finish before reading results.
 * ---------------------------------------------------------------------------------- */
typedef struct kgwas_multiscan kgwas_multiscan;
int kgwas_multiscan_create(const kgwas_scan_params* p, const int32_t* devices, uint32_t n_devices, kgwas_multiscan** out);
/* Rows [row0, row0 + n_rows) of an open .table: shard g = the g-th of n_devices equal contiguous pieces, streamed
 * through kgwas_scan_feed_table on its device. */
int kgwas_multiscan_run_table(kgwas_multiscan* m, kgwas_table* t, uint64_t row0, uint64_t n_rows);
/* Shards already resident in HBM: d_rows[g] on devices[g] holds n_rows[g] rows starting at file row first_row[g]
 * (consecutive shards, in row order). */
int kgwas_multiscan_run_device(kgwas_multiscan* m, const void* const* d_rows, const uint64_t* n_rows, const uint64_t* first_row);
int kgwas_multiscan_finish(kgwas_multiscan* m);
/* As kgwas_scan_result, for the whole table. */
int kgwas_multiscan_result(kgwas_multiscan* m, uint64_t j, uint64_t* n, const uint64_t** kmer, const double** score,
                           const uint64_t** row);
/* total: counters summed over the shards (rows_tested = .tested_kmers, patterns over all shards), times of the slowest
 * shard; per_shard [n_devices] (NULL: skip) = each session's own statistics of the last run; scan_ms / merge_ms: wall
 * time of the shard scans and of the cross-shard merges so far; rescans: shards scanned twice (ring too short). */
int kgwas_multiscan_get_stats(const kgwas_multiscan* m, kgwas_scan_stats* total, kgwas_scan_stats* per_shard, double* scan_ms,
                              double* merge_ms, uint64_t* rescans);
void kgwas_multiscan_destroy(kgwas_multiscan* m);

/* calculate_kmer_score for every row and phenotype column (src/kmers_multiple_databases.cpp:327-363):
 * scores[j*n_rows + r] (0 for rows the MAC filter drops), popcnt[r] = masked popcount N1,
 * outputs in host memory. rows_on_device selects how `rows` is interpreted. */
int kgwas_scan_scores_dense(kgwas_scan* s, const void* rows, int rows_on_device, uint64_t n_rows, double* scores,
                            uint32_t* popcnt);

/* Cross-shard merge: replay shard histories (shards in row order) through fresh heaps.
 * counts[g*n_pheno + j] entries of shard g / phenotype j, concatenated by phenotype inside
 * kmer[g] / score[g] / row[g]. Returns one heap per phenotype in out_heaps[j]. */
int kgwas_merge_shards(uint64_t n_pheno, const uint64_t* topn, uint64_t n_shards, const uint64_t* counts,
                       const uint64_t* const* kmer, const double* const* score, const uint64_t* const* row,
                       uint32_t threads, kgwas_heap** out_heaps);

/* ------------------------------------------------------------------------------------
 * Kinship = emma_kinship_kmers (src/emma_kinship_kmers.cpp:77-111) with
 * update_emma_kinshhip_calculation (src/kmers_multiple_databases.cpp:418-438):
 * over rows with min_count <= popcount(all S_f columns) <= S_f - min_count,
 * K[i][j] += 1 ^ g_i ^ g_j for j < i.
 * ---------------------------------------------------------------------------------- */
typedef struct kgwas_kinship kgwas_kinship;
int kgwas_kinship_create(int32_t device, uint64_t n_acc_file, uint64_t min_count, kgwas_kinship** out);
int kgwas_kinship_feed_device(kgwas_kinship* k, const void* d_rows, uint64_t n_rows, void* hip_stream);
int kgwas_kinship_feed_host(kgwas_kinship* k, const uint64_t* rows, uint64_t n_rows);
/* Rows [row0, row0 + n_rows) of an open .table through the double-buffered ingest (file read, copy and kernels
 * overlap); replaces the reference's load-a-batch-then-accumulate loop (src/emma_kinship_kmers.cpp:86-99). */
int kgwas_kinship_feed_table(kgwas_kinship* k, kgwas_table* t, uint64_t row0, uint64_t n_rows);
/* Hamming-distance partials (S_f x S_f u64, full symmetric) and rows used so far: integer, so
 * partials of different shards simply add (all-reduce) before kgwas_kinship_from_partials. */
int kgwas_kinship_partials(kgwas_kinship* k, uint64_t* hamming, uint64_t* n_used);
/* K (S_f x S_f, lower triangle j < i filled like the reference, rest 0) from summed partials. */
int kgwas_kinship_from_partials(uint64_t n_acc_file, const uint64_t* hamming, uint64_t n_used, uint64_t* K);
int kgwas_kinship_get_stats(const kgwas_kinship* k, double* kernel_ms, uint64_t* launches, uint64_t* rows_fed);
void kgwas_kinship_destroy(kgwas_kinship* k);
/* The whole table over several devices (one process, one thread + session per device, contiguous row shards, integer
 * partials added): hamming [S_f x S_f] and n_used as kgwas_kinship_partials gives them for a single device. */
int kgwas_kinship_table_multi(const int32_t* devices, uint32_t n_devices, kgwas_table* t, uint64_t min_count, uint64_t* hamming,
                              uint64_t* n_used);
/* The matrix text emma_kinship_kmers prints to stdout (:95-111). Returns needed bytes. */
uint64_t kgwas_kinship_format(uint64_t n_acc_file, const uint64_t* K, uint64_t n_used, char* out, uint64_t cap);

/* ------------------------------------------------------------------------------------
 * Pass 2 of associate_kmers (src/associate_kmers.cpp:150-205): <base>.bed/.bim/.fam for one
 * phenotype column from its heap (pop order) — write_PA (src/kmers_multiple_databases.cpp:218-252),
 * BedBimFilesHandle (src/kmer_general.h:133-147), write_fam_file (src/kmer_general.cpp:207-225).
 * The winners' rows are fetched by file row index instead of re-scanning the table.
 * ---------------------------------------------------------------------------------- */
int kgwas_write_plink(const char* out_base, kgwas_table* t, const uint64_t* col, uint64_t n_acc,
                      const char* const* acc_names, const float* y, uint64_t n, const uint64_t* kmer_pop,
                      const uint64_t* row_pop);
/* The same for ALL phenotype columns of a run in one call - what the loop of src/associate_kmers.cpp:167-195 writes from
 * its second pass over the table: column j's files <out_bases[j]>.bed/.bim/.fam from its heap in pop order (n_win[j]
 * entries kmer_pop[j][i], row_pop[j][i]) and its values Y[j * n_acc ..]. The union of the winners' rows is read once
 * (sorted, neighbouring rows in one read, read-ahead requested when the file turns out to be cold), expanded once
 * (word-wise) and written by `threads` host threads (0: the CPUs this process may use). Files are byte-identical to
 * kgwas_write_plink's, which is this call with one column. */
int kgwas_write_plink_many(uint64_t n_cols, const char* const* out_bases, kgwas_table* t, const uint64_t* col, uint64_t n_acc,
                           const char* const* acc_names, const float* Y, const uint64_t* n_win, const uint64_t* const* kmer_pop,
                           const uint64_t* const* row_pop, uint32_t threads);

/* ------------------------------------------------------------------------------------
 * kmers_table_to_bed (src/kmers_table_to_bed.cpp:93-129): the whole table, MAC-filtered on the phenotyped
 * accessions col[0..n_acc), as PLINK files <out_base>.<i>.bed/.bim/.fam, a new file set after every batch_size KEPT
 * k-mers (load_kmers, src/kmers_multiple_databases.cpp:110-113); with unique_patterns only the first k-mer of each
 * presence/absence hash (hash_presence_absence_pattern, :367-375) over all batches
 * (output_plink_bed_file_unique_presence_absence_patterns, :254-264). Per-row work (squeeze, popcount, hash, PLINK
 * bytes) runs on `device`. n_batches / n_written (k-mers written) may be NULL.
 * ---------------------------------------------------------------------------------- */
int kgwas_table_to_bed(kgwas_table* t, const uint64_t* col, uint64_t n_acc, const char* const* acc_names, const float* y,
                       uint64_t min_count, uint64_t batch_size, int unique_patterns, const char* out_base, int device,
                       uint64_t* n_batches, uint64_t* n_written);

/* ------------------------------------------------------------------------------------
 * SNP twin of the scorer: MultipleSNPsDataBases (src/snps_multiple_databases.h:25-63) for associate_snps
 * (src/associate_snps.cpp). open: <base>.fam/.bed with the phenotyped samples in phenotype order (ctor, :66-146,
 * error texts included); scores: calculate_grammmar_approx_association (:155-172) of every SNP for every column of
 * Y[n_pheno][n_samples], scores[j*n_snps + i], on `device`; best: get_most_associated_snps (:229-241) per column -
 * counts[j] sorted SNP indices at indices[j*topn ...]; write: output_plink_bed_file (:252-286) - list l
 * (counts[l] sorted indices at indices[l*stride ...]) to out_bases[l].bed/.bim.
 * ---------------------------------------------------------------------------------- */
typedef struct kgwas_snps kgwas_snps;
int kgwas_snps_open(const char* base_bedbim, const char* const* samples, uint64_t n_samples, kgwas_snps** out);
int kgwas_snps_info(const kgwas_snps* s, uint64_t* n_snps, uint64_t* n_samples_file, uint64_t* bytes_per_snp);
int kgwas_snps_scores(kgwas_snps* s, const float* Y, uint64_t n_pheno, double mac, int device, double* scores);
int kgwas_snps_best(kgwas_snps* s, const float* Y, uint64_t n_pheno, uint64_t topn, double mac, int device, uint64_t* counts,
                    uint64_t* indices);
int kgwas_snps_write(kgwas_snps* s, uint64_t n_lists, const char* const* out_bases, const uint64_t* counts,
                     const uint64_t* indices, uint64_t stride);
void kgwas_snps_close(kgwas_snps* s);

/* ------------------------------------------------------------------------------------
 * Seeded synthetic table rows: kmer = row + 1, per-row frequency q/256 with
 * q in [5, 250], bits from a counter-based generator, so any shard can be produced on its GPU.
 * The host variant is the bit-identical twin used to write small .table files for the CLIs.
 * ---------------------------------------------------------------------------------- */
int kgwas_synth_rows_device(void* d_rows, uint64_t first_row, uint64_t n_rows, uint64_t n_acc_file, uint64_t seed,
                            void* hip_stream);
int kgwas_synth_rows_host(uint64_t* rows, uint64_t first_row, uint64_t n_rows, uint64_t n_acc_file, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* KGWAS_H */
