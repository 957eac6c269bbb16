// scan.cpp — association-scan session: pass 1 of associate_kmers (src/associate_kmers.cpp:99-148)
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

using namespace kgwas;

namespace {

// Minimal persistent worker pool: parallel_for over phenotype columns. The caller does not take part: it
// sleeps until the workers are done, so exactly size() threads are busy (sized to the CPU quota).
class Pool {
   public:
    // cpus: optional CPU to pin worker i to (empty = leave placement to the scheduler).
    explicit Pool(unsigned n, const std::vector<std::vector<int>>& cpus = {})
        : stop_(false), gen_(0), pending_(0), n_items_(0) {
        if (n < 1) n = 1;
        for (unsigned i = 0; i < n; i++) {
            const std::vector<int> mine = i < cpus.size() ? cpus[i] : std::vector<int>();
            th_.emplace_back([this, i, mine] {
                if (!mine.empty()) {
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : mine)
                        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
                }
                loop(i);
            });
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
    }
    // The two halves of parallel_for: start() hands the items out and returns, wait() blocks until every worker is
    // done. Between the two the workers run on their own (the streaming replay: items are worker loops that end when
    // told to). fn must stay alive until wait() returns; one start at a time.
    void start(size_t n, const std::function<void(size_t)>& fn) {
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        n_items_ = n;
        if (claimed_.size() < n) claimed_ = std::vector<std::atomic<uint8_t>>(n);
        for (size_t i = 0; i < n; i++) claimed_[i].store(0, std::memory_order_relaxed);
        pending_ = th_.size();
        done_.store(0, std::memory_order_relaxed);
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
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
    size_t size() const { return th_.size(); }
    bool finished() const { return done_.load(std::memory_order_acquire) != 0; }  // every worker is through the started items

   private:
    // Own items first, in increasing order; then take whatever nobody has started yet, from the far end
    // (with 101 columns on 16 workers the five workers that own a seventh column give it away to a worker
    // that is done with its six). An item runs exactly once, on one thread.
    void run(size_t me) {
        const size_t T = th_.size();
        for (size_t i = me; i < n_items_; i += T)
            if (!claimed_[i].exchange(1, std::memory_order_acq_rel)) (*fn_)(i);
        for (size_t i = n_items_; i-- > 0;)
            if (!claimed_[i].load(std::memory_order_relaxed) && !claimed_[i].exchange(1, std::memory_order_acq_rel)) (*fn_)(i);
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
            }
            run(me);
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (--pending_ == 0) {
                    done_.store(1, std::memory_order_release);
                    done_cv_.notify_all();
                }
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    bool stop_;
    std::atomic<uint64_t> gen_;
    std::atomic<int> done_{0};
    std::vector<std::atomic<uint8_t>> claimed_;
    size_t pending_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_items_;
};

std::vector<int> parse_cpulist(const char* path) {
    std::vector<int> out;
    FILE* f = fopen(path, "r");
    if (!f) return out;
    char buf[4096];
    if (fgets(buf, sizeof(buf), f)) {
        for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            const int k = sscanf(tok, "%d-%d", &a, &b);
            if (k == 1) b = a;
            if (k >= 1)
                for (int c = a; c <= b; c++) out.push_back(c);
        }
    }
    fclose(f);
    return out;
}

// One CPU per replay worker: distinct physical cores of the NUMA node the GPU hangs off (where its mapped
// host buffers are best read), spread evenly over that node's cores (= over its L3 slices). A worker owns a
// fixed set of heaps (static assignment above), ~2 MB of state that should stay in that core's L2/L3 from
// chunk to chunk instead of following the scheduler around a 256-CPU host. Returns {} (no pinning) whenever
// the topology cannot be read or does not offer n allowed cores; KGWAS_PIN_THREADS=0 turns it off.
std::vector<std::vector<int>> pick_replay_cpus(unsigned n, int device) {
    std::vector<std::vector<int>> none;
    int mode = 1;  // 0 off, 1 one core each, 2 the GPU's share of its NUMA node for all, 3 the core's L3 domain
    if (const char* e = getenv("KGWAS_PIN_THREADS")) mode = atoi(e);
    if (mode == 0) return none;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return none;
    int node = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) == hipSuccess) {
        for (char* c = bus; *c; c++) *c = (char)tolower(*c);
        char path[256];
        snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
        if (FILE* f = fopen(path, "r")) {
            if (fscanf(f, "%d", &node) != 1) node = -1;
            fclose(f);
        }
    }
    std::vector<int> cand;
    if (node >= 0) {
        char path[256];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        cand = parse_cpulist(path);
    }
    if (cand.empty())
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) cand.push_back(c);
    std::vector<int> cores;  // first hardware thread of every allowed core
    for (int c : cand) {
        if (c < 0 || c >= CPU_SETSIZE || !CPU_ISSET(c, &allowed)) continue;
        char path[256];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        const std::vector<int> sib = parse_cpulist(path);
        if (sib.empty() || sib[0] == c) cores.push_back(c);
    }
    // Several GPUs usually share a NUMA node and each has its own process (one rank per GPU): give every
    // GPU of the node its own contiguous share of the node's cores, by PCI order, so ranks never stack.
    size_t n_gpus = 1, ordinal = 0;
    if (node >= 0 && bus[0]) {
        std::vector<std::string> gpus;
        if (DIR* d = opendir("/sys/bus/pci/devices")) {
            while (struct dirent* de = readdir(d)) {
                if (de->d_name[0] == '.') continue;
                char path[512];
                unsigned vendor = 0, cls = 0;
                int nn = -2;
                snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/vendor", de->d_name);
                if (FILE* f = fopen(path, "r")) {
                    if (fscanf(f, "%x", &vendor) != 1) vendor = 0;
                    fclose(f);
                }
                if (vendor != 0x1002) continue;
                snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/class", de->d_name);
                if (FILE* f = fopen(path, "r")) {
                    if (fscanf(f, "%x", &cls) != 1) cls = 0;
                    fclose(f);
                }
                if ((cls >> 8) != 0x0380 && (cls >> 8) != 0x1200 && (cls >> 8) != 0x0300) continue;
                snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", de->d_name);
                if (FILE* f = fopen(path, "r")) {
                    if (fscanf(f, "%d", &nn) != 1) nn = -2;
                    fclose(f);
                }
                if (nn == node) gpus.push_back(de->d_name);
            }
            closedir(d);
        }
        std::sort(gpus.begin(), gpus.end());
        for (size_t i = 0; i < gpus.size(); i++)
            if (gpus[i] == bus) {
                n_gpus = gpus.size();
                ordinal = i;
            }
    }
    const size_t share = cores.size() / n_gpus;
    if (share < n || n == 0) return none;
    std::vector<std::vector<int>> out;
    auto with_siblings = [&](int c, std::vector<int>& dst) {
        char path[256];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        std::vector<int> sib = parse_cpulist(path);
        if (sib.empty()) sib.push_back(c);
        for (int x : sib)
            if (x >= 0 && x < CPU_SETSIZE && CPU_ISSET(x, &allowed)) dst.push_back(x);
    };
    for (unsigned i = 0; i < n; i++) {
        const int core = cores[ordinal * share + (size_t)i * share / n];
        std::vector<int> set;
        if (mode == 1) {
            set.push_back(core);
        } else if (mode == 2) {
            for (size_t k = 0; k < share; k++) with_siblings(cores[ordinal * share + k], set);
        } else {
            char path[256];
            snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", core);
            for (int x : parse_cpulist(path))
                if (x >= 0 && x < CPU_SETSIZE && CPU_ISSET(x, &allowed)) set.push_back(x);
            if (set.empty()) set.push_back(core);
        }
        out.push_back(set);
    }
    if (getenv("KGWAS_TRACE")) {
        fprintf(stderr, "[kgwas] replay workers placed (mode %d, numa node %d, gpu %zu of %zu on it):", mode, node, ordinal,
                n_gpus);
        for (auto& v : out) fprintf(stderr, " %d%s", v[0], v.size() > 1 ? "+" : "");
        fprintf(stderr, "\n");
    }
    return out;
}

// CPUs this process may actually use: the cgroup CPU quota when there is one (containers often
// expose every host CPU to hardware_concurrency() while capping the quota far lower; running more
// busy threads than the quota gets the whole process throttled for the rest of the period).
unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[64];
        unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const unsigned long long quota = strtoull(q, nullptr, 10);
            if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<unsigned long long>(1, quota / period));
        }
        fclose(f);
    } else {
        long long quota = -1, period = 0;  // cgroup v1
        if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(fq, "%lld", &quota) != 1) quota = -1;
            fclose(fq);
        }
        if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(fp, "%lld", &period) != 1) period = 0;
            fclose(fp);
        }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota / period));
    }
    return n;
}

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
    // h_meta: [0, P) candidates per column, [P, 2P) their offsets, [2P] total, [2P + 1] survivor keys emitted
    double* so_score = nullptr;   // host copies: a piece of the session's pinned record ring (fetch_records)
    uint64_t* so_kmer = nullptr;
    uint32_t* so_row = nullptr;
    size_t ring_end = 0;          // ring offset behind this chunk's records (where the ring is free again once it is replayed)
    DevBuf<double> d_so_score;
    DevBuf<uint64_t> d_so_kmer;
    DevBuf<uint32_t> d_so_row;
    DevBuf<uint32_t> d_meta;
    PinBuf<uint32_t> h_meta;
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

}  // namespace

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
        bool wide = false;  // score_wide.hip: all T tiles' accumulators in registers, operands streamed through LDS
        uint32_t T = 0, n_lgroups = 0;
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
    DevBuf<uint8_t> d_Bn;
    DevBuf<NarrowCol> d_ncols;
    DevBuf<unsigned long long> d_bitmap;  // survivors of the chunk being filtered: [n_pheno][bitmap_words]
    uint64_t bitmap_words = 0;
    DevBuf<uint32_t> d_bm_blocks;
    // survivors of the chunk being filtered: bitmap [column][64-row word], then row-ordered keys per column with each
    // column's range; shared by all chunks (consumed by the re-score kernel in stream order)
    DevBuf<uint32_t> d_surv_sorted, d_surv_cnt, d_surv_off, d_key_count;  // row-ordered keys per column, the columns' ranges, the total
    DevBuf<uint32_t> d_tile_pref, d_tile_cnt, d_tile_off;  // tiles of 256 survivors (launch_rescore)
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
    size_t ring_size = 0, ring_head = 0, ring_tail = 0;  // used: [tail, head) circularly; head == tail: empty
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
    size_t n_groups = 1;
    std::vector<std::vector<uint32_t>> grp_cols;  // columns of group g (at most BestHeap::MAX_LOCKSTEP)
    std::vector<int> grp_home;                    // the worker that owns group g, -1: floating (anybody takes it)
    std::unique_ptr<GroupState[]> gstate;
    std::unique_ptr<std::atomic<uint32_t>[]> slot_left;  // [n_slots] groups that have not replayed the slot's chunk yet
    std::atomic<uint64_t> seq_submitted{0}, seq_published{0}, seq_replayed{0};
    std::atomic<bool> rp_quit{false}, rp_failed{false};
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
        if (ev_d0) (void)hipEventDestroy(ev_d0);
        if (ev_d1) (void)hipEventDestroy(ev_d1);
        if (stream) (void)hipStreamDestroy(stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

namespace {

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
    for (uint64_t j = 0; j < s->n_pheno; j++) h[j] = s->heaps[j].full() ? s->heaps[j].lowest() : 0.0;
    KGWAS_HIP(hipMemcpyAsync(s->d_thr_host.p, h, s->n_pheno * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (!s->hist_ready)
        KGWAS_HIP(hipMemcpyAsync(s->d_thr.p, h, s->n_pheno * sizeof(double), hipMemcpyHostToDevice, s->stream));
}

// First sparse chunk: histogram bin 0 of every column starts at its current exact minimum.
void start_histograms(kgwas_scan* s) {
    for (uint64_t j = 0; j < s->n_pheno; j++) {
        const double low = s->sel_valid ? s->h_sel.p[j] : s->heaps[j].lowest();
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
    for (auto& h : s->heaps) all = all && h.full();
    s->all_full = all;
}

// Dense chunk: every score comes back; replay every kept row into every heap.
void dense_fill(kgwas_scan* s, uint64_t n_rows, uint64_t first_row, std::chrono::steady_clock::time_point td0,
                const std::function<void()>* meanwhile = nullptr);

// select: also pick each column's topn-th largest score of the chunk on the device and make it the device's threshold
// (d_thr, d_thr_host) - see feed_device_impl; h_sel / h_sel_info arrive with the scores.
void run_dense(kgwas_scan* s, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row, double* out_scores,
               uint32_t* out_n1, bool replay, bool select = false) {
    ScoreArgs a;
    auto td0 = std::chrono::steady_clock::now();
    hipEvent_t e0 = s->ev_d0, e1 = s->ev_d1, es = s->ev_ds;
    fill_args(s, a, d_rows, n_rows, first_row, !s->direct);
    a.dense = s->d_dense.p;
    a.n1_out = s->d_n1.p;
    a.kmer_out = s->d_kmer.p;
    a.tested = s->d_tested_dense.p;
    KGWAS_HIP(hipMemsetAsync(s->d_tested_dense.p, 0, TESTED_SHARDS * sizeof(unsigned long long), s->stream));
    KGWAS_HIP(hipEventRecord(es, s->stream));
    maybe_squeeze(s, d_rows, n_rows);
    KGWAS_HIP(hipEventRecord(e0, s->stream));
    launch_score(s, a);
    KGWAS_HIP(hipEventRecord(e1, s->stream));
    if (select) {
        KGWAS_HIP(launch_dense_select(s->d_dense.p, s->d_n1.p, (uint32_t)n_rows, (uint32_t)s->n_pheno, (uint32_t)s->S,
                                      (uint32_t)std::min<uint64_t>(s->min_count, 0xFFFFFFFFull), s->d_topn.p, s->d_thr.p, s->d_thr_host.p,
                                      s->d_sel.p, s->d_sel_info.p, s->stream));
        KGWAS_HIP(hipMemcpyAsync(s->h_sel.p, s->d_sel.p, s->n_pheno * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        KGWAS_HIP(hipMemcpyAsync(s->h_sel_info.p, s->d_sel_info.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    }
    KGWAS_HIP(hipMemcpyAsync(s->h_dense.p, s->d_dense.p, s->n_pheno * n_rows * sizeof(double), hipMemcpyDeviceToHost,
                             s->stream));
    KGWAS_HIP(hipMemcpyAsync(s->h_n1.p, s->d_n1.p, n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    KGWAS_HIP(hipMemcpyAsync(s->h_kmer.p, s->d_kmer.p, n_rows * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    KGWAS_HIP(hipStreamSynchronize(s->stream));
    float ms = 0;
    KGWAS_HIP(hipEventElapsedTime(&ms, e0, e1));
    s->st.score_kernel_ms += ms;
    if (!s->direct) {
        KGWAS_HIP(hipEventElapsedTime(&ms, es, e0));
        s->st.squeeze_kernel_ms += ms;
    }
    s->st.chunks++;
    if (out_scores) memcpy(out_scores, s->h_dense.p, s->n_pheno * n_rows * sizeof(double));
    if (out_n1) memcpy(out_n1, s->h_n1.p, n_rows * sizeof(uint32_t));
    if (!replay || select) return;  // select: the caller pushes the rows (dense_fill) after submitting sparse chunks
    dense_fill(s, n_rows, first_row, td0);
}

// The host half of a dense chunk: every MAC-passing row into every heap. meanwhile: run by the calling thread while the
// pool's workers push (the control thread submits the first sparse chunks there).
void dense_fill(kgwas_scan* s, uint64_t n_rows, uint64_t first_row, std::chrono::steady_clock::time_point td0,
                const std::function<void()>* meanwhile) {
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
    if (meanwhile) (*meanwhile)();
    s->pool->wait();
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
    const double tiles0 = (double)s->cmode[0].tile_slices, tiles1 = (double)s->cmode[1].tile_slices;
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
        for (uint64_t j = 0; j < s->n_pheno; j++) s->h_thr_redo.p[j] = s->heaps[j].lowest();
        KGWAS_HIP(hipMemcpyAsync(s->d_thr_redo.p, s->h_thr_redo.p, s->n_pheno * sizeof(double), hipMemcpyHostToDevice,
                                 s->stream));
        a.thr = s->d_thr_redo.p;
    }
    const bool use_coarse = s->coarse && count_hist;
    KGWAS_HIP(launch_chunk_prep(sl.d_cnt.p, (uint32_t)s->n_pheno, sl.d_tested.p, use_coarse ? s->d_key_count.p : nullptr, s->stream));
    KGWAS_HIP(hipEventRecord(sl.ev_sq0, s->stream));
    maybe_squeeze(s, d_rows, n_rows);
    KGWAS_HIP(hipEventRecord(sl.ev_k0, s->stream));
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
        static const uint32_t rpb_env = getenv("KGWAS_COARSE_RPB") ? (uint32_t)atoi(getenv("KGWAS_COARSE_RPB")) : 0u;  // experiments
        // Survivors leave the filter as a bitmap [column][64-row word] of this chunk (zeroed here), which a popcount
        // scan turns into row-ordered keys per column: no key list, no sort (launch_bitmap_keys).
        const uint64_t n_words = (n_rows + 63) / 64;
        KGWAS_HIP(hipMemsetAsync(s->d_bitmap.p, 0, (size_t)s->n_pheno * n_words * 8, s->stream));
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
            // (short blocks: three 4-wave blocks share a CU and a launch's block count is rarely a multiple of the
            // 768 block slots, so long blocks leave CUs idle at the end of every launch: 4096 rows per block measured
            // 3.4 ms per 100 M rows, 768 rows 3.0)
            KGWAS_HIP(launch_narrow(na, rpb_env ? rpb_env : (n_rows >= (1u << 18) ? 768u : 256u), s->stream));
        } else {
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
                    KGWAS_HIP(launch_mx(x, Pt.T, rpb_env ? rpb_env : (n_rows >= (1u << 22) ? 4096u : n_rows >= (1u << 20) ? 2048u : 512u), s->stream));
                } else if (Pt.wide)
                    KGWAS_HIP(launch_wide(c, Pt.T, rpb_env ? rpb_env : (n_rows >= (1u << 20) ? 1024u : 256u), s->stream));
                else
                    KGWAS_HIP(launch_coarse(c, Pt.T, rpb_env ? rpb_env : (n_rows >= (1u << 22) ? 4096u : n_rows >= (1u << 20) ? 2048u : 512u), s->stream));
            }
        }
        KGWAS_HIP(hipEventRecord(sl.ev_mid, s->stream));
        a.tested = nullptr;  // counted by the filter
        KGWAS_HIP(launch_bitmap_keys(s->d_bitmap.p, n_words, n_rows, (uint32_t)s->n_pheno, s->d_bm_blocks.p, s->d_surv_sorted.p, s->key_slots,
                                     s->row_key_bits, s->d_surv_off.p, s->d_surv_cnt.p, s->d_key_count.p, s->d_tile_pref.p, /*nibble_transposed=*/!s->narrow, s->stream));
        a.so_score = sl.d_so_score.p;
        a.so_kmer = sl.d_so_kmer.p;
        a.so_row = sl.d_so_row.p;
        KGWAS_HIP(launch_rescore(a, s->d_surv_sorted.p, s->d_surv_off.p, s->d_surv_cnt.p, s->row_key_bits, s->d_tile_pref.p,
                                 s->d_tile_cnt.p, s->d_tile_off.p, s->d_tmp_score.p, s->d_key_count.p, sl.d_meta.p, s->stream));
        s->st.score_launches++;
    } else {
        launch_score(s, a);
    }
    KGWAS_HIP(hipEventRecord(sl.ev_k1, s->stream));
    if (s->hist_ready)  // raise the thresholds for whatever is queued next; no host round trip
        KGWAS_HIP(launch_thr_update(s->d_hist.p, s->d_hist_base.p, HIST_BINS, s->d_topn.p, s->d_thr_host.p, s->d_thr.p,
                                    (uint32_t)s->n_pheno, s->stream));
    KGWAS_HIP(hipMemcpyAsync(sl.h_tested.p, sl.d_tested.p, TESTED_SHARDS * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                             s->stream));
    if (use_coarse) {
        // the record copies follow on the copy stream once the control thread has read the counts (fetch_records)
        KGWAS_HIP(hipMemcpyAsync(sl.h_meta.p, sl.d_meta.p, (2 * s->n_pheno + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
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

void process_range_sync(kgwas_scan* s, Slot& sl, const uint64_t* d_rows, uint64_t n_rows, uint64_t first_row);

// ---- replay of a sparse chunk's records -------------------------------------------------------------------
// The unit of host work is (chunk, column group): group g owns the columns g, g + n_groups, ... and replays a
// chunk's records of those columns in row order. A group's chunks must be replayed in submission order; different
// groups are independent (columns never interact). Two drivers use it: the streaming replay below (workers pick
// (chunk, group) units as the GPU completes chunks, no barrier between chunks) and the synchronous overflow path.
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
        for (const uint32_t j : members) {
            const uint32_t n = sl.h_meta.p[j];
            if (!n) continue;
            const uint64_t o = sl.h_meta.p[s->n_pheno + j];
            cols[n_cols++] = Cur{sl.so_score + o, sl.so_kmer + o, sl.so_row + o, 0, n, &s->heaps[j], j};
        }
        // The records were just written by the GPU (no CPU cache holds them) and the replay walks several short
        // streams at once, more than the hardware prefetchers track: pull them in up front, a line at a time.
        // (Prefetching ALL of a unit's records up front - up to a megabyte - pushed the heaps out of the L2 and the
        // records' own first lines out again before their turn: 10 % of the replay's CPU time.)
        static const uint32_t PF_AHEAD = getenv("KGWAS_REPLAY_PF") ? (uint32_t)atoi(getenv("KGWAS_REPLAY_PF")) : 64u;  // records
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
                    if (open || v > low) {  // (-inf, "not a candidate", never gets here: the device ships candidates only)
                        ready = v != none;
                        if (ready) break;
                        advance(cu);
                        continue;
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
            const bool full = h.full();
            const double low = h.lowest();
            for (uint32_t i = 0; i < n; i++)
                if (!full || c[i].score > low) keys.push_back(((c[i].row - row0) << 32) | i);
            std::sort(keys.begin(), keys.end());
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
    for (uint32_t i = 0; i < TESTED_SHARDS; i++) s->st.rows_tested += sl.h_tested.p[i];
    return true;
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
    std::vector<ReplayAcc> accs(s->n_groups);
    s->pool->parallel_for(s->n_groups, [&](size_t g) { replay_group(s, sl, g, accs[g]); });
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

uint64_t next_sparse_chunk(const kgwas_scan* s) {
    // The device keeps its thresholds current with everything submitted so far (thr_update_kernel), so a
    // chunk of c rows ships about topn * c / rows_submitted records per column. Exact scorer: each column's list
    // holds cap records, keep that under cap / 3. Int8 filters: the survivor keys of all columns share one list of
    // key_slots (and so do the records); plan for half of it with the survivors per candidate that the finished
    // chunks of the coming chunk's mode reported (~1 for the narrow filter).
    const double m = (double)std::max<uint64_t>(s->rows_submitted, 1);
    static const double fill = getenv("KGWAS_FILL") ? atof(getenv("KGWAS_FILL")) : 0.4;  // experiments
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
    for (int i = 0; i < 400 && !done; i++) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) done = true;
        else if (q != hipErrorNotReady) KGWAS_HIP(q);
        else
            for (int k = 0; k < 20; k++) __builtin_ia32_pause();
    }
    if (!done) KGWAS_HIP(hipEventSynchronize(ev));
    s->st.gpu_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
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
        while (s->ring_freed < rep) {
            s->ring_tail = s->slot[(size_t)(s->ring_freed % (uint64_t)s->n_slots)].ring_end;
            s->ring_freed++;
        }
        // empty: start over at the bottom - but only when every chunk fetched before this one has been freed: a fetched,
        // not yet replayed chunk without records (or one that overflowed) carries a ring_end taken from the old head,
        // and freeing it later would move the tail back over records placed at the bottom in the meantime
        if (s->ring_tail == s->ring_head && s->ring_freed == seq) s->ring_head = s->ring_tail = 0;
    }
    if (copy) {
        const size_t need = ((size_t)n * 20 + 63) / 64 * 64;
        size_t at;
        if (s->ring_head >= s->ring_tail) {  // used part does not wrap (or the ring is empty)
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
    if (copy) {
        KGWAS_HIP(hipMemcpyAsync(sl.so_score, sl.d_so_score.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        KGWAS_HIP(hipMemcpyAsync(sl.so_row, sl.d_so_row.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->copy_stream));
        KGWAS_HIP(hipMemcpyAsync(sl.so_kmer, sl.d_so_kmer.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, s->copy_stream));
    }
    KGWAS_HIP(hipEventRecord(sl.ev_done, s->copy_stream));
    return true;
}

// ---- streaming replay ----------------------------------------------------------------------------------------
// Sparse chunks carry sequence numbers (slot = seq % n_slots). The control thread (the caller of a feed) submits
// chunks while slots are free, waits for the GPU to finish them in order and PUBLISHES them (seq_published); the
// pool's workers, running one long item each, pick (chunk, group) units: any group that is not being worked on and
// whose next chunk is published, the group furthest behind first. There is no barrier between chunks, so a chunk's
// slowest group does not hold the others up (the per-chunk barrier cost 17 % of the replay at 101 columns on 16
// workers), and the pool is woken once per feed instead of once per chunk. A slot is reused when all groups have
// replayed its chunk (seq_replayed counts such chunks; they complete in order).
void replay_worker(kgwas_scan* s, size_t w) {
    ReplayAcc acc;
    const size_t NG = s->n_groups;
    int idle_spins = 0;
    try {
        for (;;) {
            if (s->rp_quit.load(std::memory_order_acquire)) break;
            const uint64_t pub = s->seq_published.load(std::memory_order_acquire);
            // The group furthest behind among this worker's own groups and the floating ones; another worker's group
            // only when that worker has fallen two published chunks behind (a heap that changes cores drags its
            // 320 KB along, so groups stay at home unless a core is really slow - a co-tenant, a throttled sibling).
            size_t best = (size_t)-1;
            uint64_t best_done = ~0ull;
            for (int pass = 0; pass < 2 && best == (size_t)-1; pass++)
                for (size_t g = 0; g < NG; g++) {
                    const int home = s->grp_home[g];
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
                const size_t si = (size_t)(d % (uint64_t)s->n_slots);
                const double tr0 = s->trace ? s->t_ms() : 0.0;
                replay_group(s, s->slot[si], best, acc);
                if (s->trace && (s->n_groups <= 4 || best == 0))
                    fprintf(stderr, "[kgwas t=%.3f] worker %zu replayed chunk %llu group %zu in %.3f ms\n", s->t_ms(), w, (unsigned long long)d, best, s->t_ms() - tr0);
                G.done.store(d + 1, std::memory_order_release);
                G.busy.store(0u, std::memory_order_release);
                idle_spins = 0;
                if (s->slot_left[si].fetch_sub(1u, std::memory_order_acq_rel) == 1u) {  // the chunk's last group
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
            if (++idle_spins < 64) {
                for (int i = 0; i < 32; i++) __builtin_ia32_pause();
                continue;
            }
            std::unique_lock<std::mutex> lk(s->rp_mu);
            if (s->rp_quit.load(std::memory_order_acquire)) break;
            s->rp_idle.fetch_add(1, std::memory_order_relaxed);
            s->rp_cv_work.wait_for(lk, std::chrono::microseconds(200));
            s->rp_idle.fetch_sub(1, std::memory_order_relaxed);
        }
    } catch (...) {
        s->rp_failed.store(true, std::memory_order_release);
    }
    std::lock_guard<std::mutex> lk(s->rp_mu);
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
    bool running = false;
    std::chrono::steady_clock::time_point t_start;
    auto replayed = [&]() { return s->seq_replayed.load(std::memory_order_acquire); };
    auto start_async = [&]() {
        if (running) return;
        s->rp_quit.store(false, std::memory_order_release);
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
    s->ring_head = s->ring_tail = 0;
    s->ring_freed = 0;
    for (size_t g = 0; g < s->n_groups; g++) {
        s->gstate[g].done.store(0);
        s->gstate[g].busy.store(0);
    }
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
                static const bool no_overlap = getenv("KGWAS_NO_DENSE_OVERLAP") != nullptr;  // experiments
                bool empty = s->coarse && !no_overlap && n_rows - pos > c && sub == 0;
                for (uint64_t j = 0; j < s->n_pheno && empty; j++) empty = s->heaps[j].size() == 0;
                if (empty) {
                    const auto td0 = std::chrono::steady_clock::now();
                    run_dense(s, d_rows + pos * stride, c, first_row + pos, nullptr, nullptr, true, /*select=*/true);
                    const uint64_t dense_first = first_row + pos;
                    pos += c;
                    // every heap will be full after these rows, and no score is a NaN (the heap's order with NaNs in it is
                    // not a total one: plain path)
                    const bool early = s->h_sel_info.p[0] >= s->max_topn && s->h_sel_info.p[1] == 0;
                    const std::function<void()> presubmit = [&]() {  // (the replay workers start after the fill)
                        s->sel_valid = true;
                        s->rows_submitted = std::max(s->rows_submitted, s->rows_done + c);
                        // GPU work for the duration of the fill: at least two chunks, more only while the workers are
                        // still pushing (a submission costs this thread 0.1-0.2 ms; the replay starts when both are done)
                        while (pos < n_rows && sub < std::min<uint64_t>(depth, 12) && (sub < 2 || !s->pool->finished())) {
                            const uint64_t cs = std::min<uint64_t>(next_sparse_chunk(s), n_rows - pos);
                            const size_t si = (size_t)(sub % (uint64_t)s->n_slots);
                            s->slot_left[si].store((uint32_t)s->n_groups, std::memory_order_release);
                            submit_sparse(s, s->slot[si], d_rows + pos * stride, cs, first_row + pos, /*count_hist=*/true);
                            s->rows_submitted += cs;
                            sub++;
                            s->seq_submitted.store(sub, std::memory_order_release);
                            pos += cs;
                        }
                        s->sel_valid = false;
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
                s->slot_left[si].store((uint32_t)s->n_groups, std::memory_order_release);
                submit_sparse(s, s->slot[si], d_rows + pos * stride, c, first_row + pos, /*count_hist=*/true);
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
            if (cpy < sub && ((cpy == pub && !can_submit) || hipEventQuery(s->slot[(size_t)(cpy % (uint64_t)s->n_slots)].ev_counts) == hipSuccess ||
                              !s->slot[(size_t)(cpy % (uint64_t)s->n_slots)].used_coarse)) {
                if (fetch_records(s, s->slot[(size_t)(cpy % (uint64_t)s->n_slots)], cpy)) {
                    if (s->trace) fprintf(stderr, "[kgwas t=%.3f] counts of chunk %llu in, record copy ordered\n", s->t_ms(), (unsigned long long)cpy);
                    cpy++;
                    continue;
                }
                // the record ring is full: publish what is fetched; if all of that is published, wait for the replay
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
                    for (size_t g = 0; g < s->n_groups; g++) s->gstate[g].done.store(pub, std::memory_order_release);
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

void check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error(KGWAS_ERR_DEVICE,
                    "no HIP device available: libkgwas has no CPU fallback (hipGetDeviceCount: " +
                        std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
    if (device < 0 || device >= n) throw Error(KGWAS_ERR_ARG, "device ordinal out of range");
}

}  // namespace

namespace kgwas {
unsigned usable_cpus_quota() { return usable_cpus(); }
// multiscan.cpp: the pattern hashes this session has collected so far (for the distinct count over all shards)
void scan_patterns_peek(kgwas_scan* s, const uint64_t** d_hashes, uint64_t* n, int* device) {
    KGWAS_HIP(hipSetDevice(s->device));
    KGWAS_HIP(hipStreamSynchronize(s->stream));
    unsigned long long c = 0;
    if (s->count_patterns) KGWAS_HIP(hipMemcpy(&c, s->d_pat_cnt.p, 8, hipMemcpyDeviceToHost));
    *d_hashes = s->d_pat.p;
    *n = c;
    *device = s->device;
}
// (Re)create the session's empty heaps. Their entry and payload arrays are carved out of one huge-page arena when the
// heap sizes allow it (up to 2 GiB in all), each reserved in full; larger requests grow on the ordinary heap as before.
void make_heaps(kgwas_scan* s) {
    s->heaps.clear();
    uint64_t need = 4096;
    for (uint64_t j = 0; j < s->n_pheno; j++) need += (uint64_t)s->topn[j] * 32 + 512;
    std::pmr::memory_resource* mr = nullptr;
    static const bool no_huge = getenv("KGWAS_NO_HUGE_HEAPS") != nullptr;  // experiments
    if (need <= (2ull << 30) && !no_huge) {
        const size_t bytes = (size_t)((need + (2u << 20) - 1) / (2u << 20) * (2u << 20));
        if (!s->heap_arena.p) {
            s->heap_arena.p = aligned_alloc(2u << 20, bytes);
            if (s->heap_arena.p) {
                s->heap_arena.bytes = bytes;
                (void)madvise(s->heap_arena.p, bytes, MADV_HUGEPAGE);
            }
        }
        if (s->heap_arena.p) {
            s->heap_mr.reset(new std::pmr::monotonic_buffer_resource(s->heap_arena.p, s->heap_arena.bytes, std::pmr::new_delete_resource()));
            mr = s->heap_mr.get();
        }
    }
    s->heaps.reserve(s->n_pheno);
    for (uint64_t j = 0; j < s->n_pheno; j++) {
        s->heaps.emplace_back((size_t)s->topn[j], mr);
        if (s->history_ring) s->heaps.back().enable_ring(ring_size(s->history_ring, s->topn[j]));
    }
}

}  // namespace kgwas

extern "C" {

uint32_t kgwas_host_cpu_quota(void) { return usable_cpus(); }

int kgwas_device_count(int* n_devices) {
    return guarded([&] {
        if (!n_devices) throw Error(KGWAS_ERR_ARG, "null argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *n_devices = (e == hipSuccess) ? n : 0;
    });
}

int kgwas_scan_create(const kgwas_scan_params* p, kgwas_scan** out) {
    return guarded([&] {
        if (!p || !out) throw Error(KGWAS_ERR_ARG, "kgwas_scan_create: null argument");
        if (p->struct_size != sizeof(kgwas_scan_params)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_params: size mismatch");
        if (!p->col || !p->Y || !p->topn || p->n_acc == 0 || p->n_pheno == 0 || p->n_acc_file == 0)
            throw Error(KGWAS_ERR_ARG, "kgwas_scan_create: empty problem");
        if (p->n_acc > p->n_acc_file) throw Error(KGWAS_ERR_ARG, "more phenotyped accessions than table columns");
        if (p->n_acc_file >= (1ull << 31)) throw Error(KGWAS_ERR_ARG, "too many accessions");
        check_device(p->device);
        KGWAS_HIP(hipSetDevice(p->device));
        std::unique_ptr<kgwas_scan> s(new kgwas_scan);
        s->device = p->device;
        s->S_f = p->n_acc_file;
        s->S = p->n_acc;
        s->W_f = (s->S_f + 63) / 64;
        s->W_m = 2 * ((s->S + 127) / 128);  // src/kmers_multiple_databases.cpp:51
        s->L = 64 * s->W_m;
        s->n_pheno = p->n_pheno;
        s->min_count = p->min_count;
        s->col.assign(p->col, p->col + s->S);
        s->topn.assign(p->topn, p->topn + s->n_pheno);
        s->Y.assign(p->Y, p->Y + s->n_pheno * s->S);
        if (p->record_history > 2) throw Error(KGWAS_ERR_ARG, "record_history: 0 (off), 1 (full log) or 2 (eviction ring)");
        s->record_history = p->record_history == 1;
        if (p->record_history == 2) {
            s->history_ring = 1;  // per heap: 16 standard deviations of the rank distance between two shards' N-th scores
            if (const char* e = getenv("KGWAS_HISTORY_RING"))
                if (atoll(e) > 0) s->history_ring = (size_t)atoll(e);
        }
        s->count_patterns = p->count_patterns != 0;
        std::vector<bool> seen(s->S_f, false);
        for (uint64_t i = 0; i < s->S; i++) {
            if (s->col[i] >= s->S_f) throw Error(KGWAS_ERR_ARG, "column index out of range");
            if (seen[s->col[i]]) throw Error(KGWAS_ERR_ARG, "duplicate column index");
            seen[s->col[i]] = true;
        }
        for (uint64_t j = 0; j < s->n_pheno; j++) {
            if (s->topn[j] == 0) throw Error(KGWAS_ERR_ARG, "heap size must be >= 1");
            s->max_topn = std::max(s->max_topn, s->topn[j]);
            s->sum_topn += s->topn[j];
        }
        s->direct = true;
        for (uint64_t i = 0; i < s->S; i++) s->direct = s->direct && (s->col[i] == i);

        bool finite = true;
        for (float v : s->Y) finite = finite && std::isfinite(v);
        uint32_t kern = p->kernel;
        const bool mfma_fits = mfma_lds_bytes((uint32_t)s->W_m) <= 160u * 1024u;
        // Coarse int8 filter + exact re-scoring for the sparse phase (score_coarse.hip): needs finite values,
        // an exact kernel for the dense phase / re-runs, and T >= 2 int8 tiles of the whole sample axis in LDS.
        const uint32_t n_kgroups = (uint32_t)((s->W_m + 7) / 8);
        uint32_t coarse_T = 0;
        for (uint32_t T : {8u, 7u, 6u, 5u, 4u, 3u, 2u})
            if (coarse_lds_bytes(n_kgroups, T) <= 152u * 1024u) {
                coarse_T = T;
                break;
            }
        bool want_coarse = false;
        if (kern == KGWAS_KERNEL_COARSE) {
            if (!finite || !coarse_T) throw Error(KGWAS_ERR_ARG, "coarse filter needs finite phenotype values and <= 5120 accessions");
            want_coarse = true;
            kern = KGWAS_KERNEL_AUTO;
        } else if (kern == KGWAS_KERNEL_AUTO && finite && coarse_T) {
            // any number of columns: even a single column (one mostly empty 16-column tile) runs twice as fast behind
            // the filter as through the exact VALU scorer (12.5 vs 27 ms per 100 M-row pass)
            want_coarse = true;
        }
        if (kern == KGWAS_KERNEL_AUTO) kern = (s->n_pheno >= 4 && finite && mfma_fits) ? KGWAS_KERNEL_MFMA : KGWAS_KERNEL_VALU;
        if (kern == KGWAS_KERNEL_MFMA && !mfma_fits)
            throw Error(KGWAS_ERR_ARG, "MFMA scorer: phenotype tile does not fit LDS for this many accessions");
        if (kern == KGWAS_KERNEL_MFMA && !finite)
            throw Error(KGWAS_ERR_ARG, "MFMA scorer needs finite phenotype values (0*inf); use the VALU scorer");
        if (kern != KGWAS_KERNEL_MFMA && kern != KGWAS_KERNEL_VALU) throw Error(KGWAS_ERR_ARG, "unknown kernel id");
        s->kernel_used = kern;
        s->coarse = want_coarse;
        s->coarse_T = coarse_T;
        s->n_kgroups = n_kgroups;
        // One to four columns under AUTO: the narrow filter (FP4 x FP8 block-scaled MFMA, three slices per column)
        // instead of the int8 one, whose 16-column tiles would be mostly padding (KGWAS_NARROW=0: keep the int8 filter).
        s->narrow = want_coarse && p->kernel == KGWAS_KERNEL_AUTO && s->n_pheno <= NARROW_MAX_COLS &&
                    narrow_lds_bytes(n_kgroups) <= 64u * 1024u && !(getenv("KGWAS_NARROW") && atoi(getenv("KGWAS_NARROW")) == 0);

        // (narrow filter on rows read in place: chunks of up to 32 M rows - with one column a chunk's fixed costs, sort,
        // re-score, threshold update, weigh more than the candidates a staler threshold lets through)
        s->chunk_max = p->chunk_rows ? p->chunk_rows : ((s->narrow && s->direct) ? (32ull << 20) : (8ull << 20));
        s->chunk_max = std::max<uint64_t>(128, (s->chunk_max + 127) / 128 * 128);
        if (s->coarse) {  // survivor keys are (column << row_bits | row) in 32 bits, the 0xFFFFFFFF fill included
            uint32_t pbits = 1;
            while ((1ull << pbits) < s->n_pheno + 1) pbits++;
            if (pbits > 22) throw Error(KGWAS_ERR_ARG, "coarse filter: too many phenotype columns for 32-bit survivor keys");
            s->chunk_max = std::max<uint64_t>(128, std::min<uint64_t>(s->chunk_max, 1ull << (32 - pbits)));
        }
        if (s->coarse && !s->narrow) {  // the coarse kernel addresses a chunk's rows with 32-bit byte offsets
            const uint64_t stride_dw = 2 * (1 + std::max<uint64_t>(s->W_f, s->W_m));
            const uint64_t lim = ((1ull << 32) - (1ull << 20)) / (4 * stride_dw) / 128 * 128;
            s->chunk_max = std::max<uint64_t>(128, std::min<uint64_t>(s->chunk_max, lim));
        }
        s->dense_rows = std::min<uint64_t>(16384, s->chunk_max);
        // Dense chunks of a feed: enough rows to fill the largest heap with a margin for the MAC filter (more
        // dense chunks follow while a heap is still short); everything after goes through the sparse path.
        s->dense_chunk = std::min<uint64_t>(s->dense_rows, std::max<uint64_t>(1024, (s->max_topn + s->max_topn / 8 + 512 + 127) / 128 * 128));
        if (getenv("KGWAS_MODE_K")) s->mode_k = atof(getenv("KGWAS_MODE_K"));  // experiments
        const uint64_t budget = getenv("KGWAS_CAP_BUDGET") ? strtoull(getenv("KGWAS_CAP_BUDGET"), nullptr, 10) : (4ull << 20);  // candidate records per slot
        // (few columns: longer lists, so that the ramp takes ~6 chunks instead of ~13 - a chunk's fixed costs, not its
        // rows, are what a one-column scan pays for)
        const uint64_t cap_mult = getenv("KGWAS_CAP_MULT") ? strtoull(getenv("KGWAS_CAP_MULT"), nullptr, 10) : (s->narrow ? 16 : 2);  // experiments
        uint64_t cap = std::min<uint64_t>(cap_mult * s->max_topn + 4096, std::max<uint64_t>(budget / s->n_pheno, 1024));
        s->cap = (uint32_t)std::min<uint64_t>(cap, 0x7FFFFFFFull);

        KGWAS_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        KGWAS_HIP(hipEventCreate(&s->ev_user));
        KGWAS_HIP(hipEventCreate(&s->ev_ds));
        KGWAS_HIP(hipEventCreate(&s->ev_d0));
        KGWAS_HIP(hipEventCreate(&s->ev_d1));

        // ---- constant device data --------------------------------------------------------
        const uint64_t S = s->S, L = s->L, W_m = s->W_m, P = s->n_pheno;
        std::vector<uint32_t> dmask(2 * W_m, 0), colmap(L, 0xFFFFFFFFu);
        for (uint64_t d = 0; d < 2 * W_m; d++) {
            if (!s->direct)
                dmask[d] = 0xFFFFFFFFu;
            else if (32 * d + 32 <= S)
                dmask[d] = 0xFFFFFFFFu;
            else if (32 * d < S)
                dmask[d] = (1u << (S - 32 * d)) - 1u;
        }
        for (uint64_t i = 0; i < S; i++) colmap[i] = (uint32_t)s->col[i];
        const uint64_t P4 = (P + 3) / 4 * 4, nct = (P + 15) / 16;
        std::vector<float> Yperm(P4 * L, 0.0f), Ymfma(nct * L * 16, 0.0f), sums(P, 0.0f);
        std::vector<float> V(L);
        for (uint64_t j = 0; j < P; j++) {
            std::fill(V.begin(), V.end(), 0.0f);
            for (uint64_t i = 0; i < S; i++) V[i] = s->Y[j * S + i];
            float* R = &Yperm[j * L];
            // permute_scores (src/kmer_general.cpp:155-167): R[128b+4s+l] = V[128b+32l+31-s]
            for (uint64_t b = 0; b < L / 128; b++)
                for (uint64_t sx = 0; sx < 32; sx++)
                    for (uint64_t l = 0; l < 4; l++) R[128 * b + 4 * sx + l] = V[128 * b + 32 * l + 31 - sx];
            // update_scores_and_sum (src/kmers_multiple_databases.cpp:288-295): sequential float32 sum
            volatile float sum = 0.0f;
            for (uint64_t i = 0; i < L; i++) sum = sum + R[i];
            sums[j] = sum;
            // MFMA layout (see score_mfma.hip): [ct][(((b*4+l)*2 + t/4)*64 + kk*16+n)*4 + t%4], s = 4t+kk
            const uint64_t ct = j / 16, n = j % 16;
            for (uint64_t b = 0; b < L / 128; b++)
                for (uint64_t l = 0; l < 4; l++)
                    for (uint64_t sx = 0; sx < 32; sx++)
                    {
                        // chain step sx = 4t + kk; lane = kk*16 + n; four consecutive t sit together
                        const uint64_t t = sx / 4, kk = sx % 4;
                        Ymfma[ct * L * 16 + (((b * 4 + l) * 2 + t / 4) * 64 + kk * 16 + n) * 4 + t % 4] =
                            V[128 * b + 32 * l + 31 - sx];
                    }
        }
        {
            const uint64_t avail = s->direct ? 2 * s->W_f : 2 * W_m;
            uint32_t nb = 0;
            while (nb < W_m / 2 && 4ull * nb + 3 < avail && dmask[4 * nb] == 0xFFFFFFFFu &&
                   dmask[4 * nb + 1] == 0xFFFFFFFFu && dmask[4 * nb + 2] == 0xFFFFFFFFu && dmask[4 * nb + 3] == 0xFFFFFFFFu)
                nb++;
            s->nb_full = nb;
        }
        s->d_dmask.alloc(dmask.size());
        s->d_colmap.alloc(colmap.size());
        s->d_sums.alloc(P);
        s->d_thr.alloc(P);
        s->h_thr.alloc(8 * P);
        s->d_thr_host.alloc(P);
        s->d_thr_redo.alloc(P);
        s->h_thr_redo.alloc(P);
        s->d_hist.alloc(P * (size_t)HIST_BINS);
        s->d_hist_base.alloc(P);
        s->h_hist_base.alloc(P);
        s->d_pat_cnt.alloc(1);
        KGWAS_HIP(hipMemset(s->d_pat_cnt.p, 0, 8));
        KGWAS_HIP(hipMemset(s->d_thr.p, 0, P * sizeof(double)));  // 0 = "nothing is filtered" until the heaps say otherwise
        KGWAS_HIP(hipMemset(s->d_thr_host.p, 0, P * sizeof(double)));
        s->d_topn.alloc(P);
        s->d_sel.alloc(P);
        s->h_sel.alloc(P);
        s->d_sel_info.alloc(2);
        s->h_sel_info.alloc(2);
        KGWAS_HIP(hipMemcpy(s->d_topn.p, s->topn.data(), P * 8, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(s->d_dmask.p, dmask.data(), dmask.size() * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(s->d_colmap.p, colmap.data(), colmap.size() * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(s->d_sums.p, sums.data(), P * 4, hipMemcpyHostToDevice));
        if (kern == KGWAS_KERNEL_MFMA) {
            s->d_Ymfma.alloc(Ymfma.size());
            KGWAS_HIP(hipMemcpy(s->d_Ymfma.p, Ymfma.data(), Ymfma.size() * 4, hipMemcpyHostToDevice));
        }
        if (kern != KGWAS_KERNEL_MFMA || s->coarse) {
            s->d_Yperm.alloc(Yperm.size());
            KGWAS_HIP(hipMemcpy(s->d_Yperm.p, Yperm.data(), Yperm.size() * 4, hipMemcpyHostToDevice));
        }
        if (s->coarse) {
            // int8 slices per column: y_i ~ c + u*(254*q0_i + q1_i) (two slices, ~15 bits) or c + u*q0_i (one),
            // centred at c = sum/N, sum being the reference's float32 sum of the column: then
            //   r_c = N*yc - N1*sum = N*u*Dc + N1*(N*c - sum),   |N1*(N*c - sum)| <= rho  (rounding of c only),
            // i.e. an exact integer Dc times a constant. For every row
            //   |yigi_ref - yc| <= Eg + |sum_{i in row} resid_i| <= Eg + min(Rall, N1 * rmax):
            //   Eg   = gamma_{L/4+3} * sum|y_i|  float32 summation error of the reference chains (Higham, recursive sums)
            //   Rall = max(sum of the positive resid_i, sum of the |negative resid_i|)  (a row's residuals cannot
            //          add up to more than all residuals of one sign), rmax = max_i |resid_i|,
            //          resid_i = y_i - c - u*(254 q0_i + q1_i)
            // so score_ref > thr needs (N*u*|Dc| + rho + N*E)^2 >= thr*d*(1 - 2^-40), i.e.
            //   |Dc| >= sqrt(thr)*kalpha*sqrt(d) - eg - min(rall, N1*rmax)       (units of u; score_coarse.hip)
            // with kalpha rounded down by 2^-19 relative and the error terms rounded up and padded: the device
            // evaluates the right-hand side in float32, and these margins dominate its rounding.
            const double u32 = std::ldexp(1.0, -24);
            const double nterms = (double)L / 4.0 + 3.0;
            const double gamma = nterms * u32 / (1.0 - nterms * u32);
            std::vector<int> q0(S), q1(S);
            auto up = [](double x) { return std::nextafter((float)x, std::numeric_limits<float>::infinity()); };
            struct ErrBound {
                float eg, rall, rmax;     // phenotype units, rounded up
                float egD, rallD, rmaxD;  // the same in units of Dc (divided by u), rounded up: what the kernel uses
            };
            auto quantise = [&](uint64_t j, int ns, CoarseCol& cc, ErrBound& eb) {
                const double Nd = (double)S, sum = (double)sums[j];
                const double c = sum / Nd;
                double mx = 0, A = 0;
                for (uint64_t i = 0; i < S; i++) {
                    const double y = (double)s->Y[j * S + i];
                    mx = std::max(mx, std::fabs(y - c));
                    A += std::fabs(y);
                }
                // unit u: one slice spans +-127 u, two slices +-(127*254 + 127) u
                const double u = mx > 0 ? (ns == 2 ? mx / (127.0 * 254.0) : mx / 127.0) : 1.0;
                const double a0 = ns == 2 ? 254.0 * u : u;
                double rpos = 0, rneg = 0, rmax = 0;
                for (uint64_t i = 0; i < S; i++) {
                    const double y = (double)s->Y[j * S + i] - c;
                    int v0 = (int)std::lrint(y / a0);
                    v0 = std::max(-127, std::min(127, v0));
                    double r = y - a0 * v0;
                    int v1 = 0;
                    if (ns == 2) {
                        v1 = (int)std::lrint(r / u);
                        v1 = std::max(-127, std::min(127, v1));
                        r -= u * v1;
                    }
                    q0[i] = v0;
                    q1[i] = v1;
                    if (r > 0) rpos += r; else rneg -= r;
                    rmax = std::max(rmax, std::fabs(r));
                }
                const double rho = Nd * std::fabs(Nd * c - sum) * 2.0 + 1e-9 * (1.0 + std::fabs(sum));
                const double Eg = gamma * A * (1.0 + 1e-6) + 1e-12 * (1.0 + A);
                cc.kalpha = (1.0 - std::ldexp(1.0, -19)) / (Nd * u);
                cc.iu = up(1.0 / u * (1.0 + 1e-6));
                eb.eg = up((Eg + rho / Nd) * (1.0 + 1e-6) + 1e-30);
                eb.rall = up(std::max(rpos, rneg) * (1.0 + 1e-6));
                eb.rmax = up(rmax * (1.0 + 1e-6));
                const double iu = 1.0 / u * (1.0 + 1e-6);
                eb.egD = up((double)eb.eg * iu);
                eb.rallD = up((double)eb.rall * iu);
                eb.rmaxD = up((double)eb.rmax * iu);
            };
            // One slice halves the matrix work but widens the bound; it is offered when, for every column, the bound
            // at N1 = S/2 stays below 15 % of the deviation of yigi a z = 4 association needs (2*sigma*sqrt(S)), so the
            // survivors stay within a small multiple of the true candidates. KGWAS_COARSE_SLICES=1|2 forces one set.
            bool one_ok = true;
            for (uint64_t j = 0; j < P && one_ok; j++) {
                CoarseCol cc;
                ErrBound eb;
                quantise(j, 1, cc, eb);
                double mean = 0, var = 0;
                for (uint64_t i = 0; i < S; i++) mean += (double)s->Y[j * S + i];
                mean /= (double)S;
                for (uint64_t i = 0; i < S; i++) var += ((double)s->Y[j * S + i] - mean) * ((double)s->Y[j * S + i] - mean);
                const double sigma = std::sqrt(var / (double)S);
                const double e_half = (double)eb.eg + std::min((double)eb.rall, 0.5 * (double)S * (double)eb.rmax);
                if (!(e_half <= 0.15 * 2.0 * sigma * std::sqrt((double)S))) one_ok = false;
            }
            bool want[2] = {one_ok, true};
            if (const char* e = getenv("KGWAS_COARSE_SLICES")) {
                if (atoi(e) == 1) want[0] = true, want[1] = false;
                if (atoi(e) == 2) want[0] = false, want[1] = true;
            }
            if (s->narrow) {
                want[0] = want[1] = false;  // the int8 operand sets are not needed
                // FP8 E4M3 operands of the narrow filter (score_narrow.hip): three slices of integers in [-15, 15] per
                // column, y_i - c ~ sum_k u_k q_ki with u_0 = max|y_i - c| / 15 and u_{k+1} = u_k / 30 (a rounding
                // residual of at most u_k / 2 fills the next slice's range exactly), and a ones row per column.
                auto e4m3 = [](int v) -> uint8_t {
                    if (v == 0) return 0;
                    const int sg = v < 0 ? 0x80 : 0, av = std::abs(v);
                    int e = 0;
                    while ((2 << e) <= av) e++;
                    return (uint8_t)(sg | ((e + 7) << 3) | ((av * 8) / (1 << e) - 8));
                };
                const uint64_t n_steps = 4ull * n_kgroups;
                std::vector<uint8_t> Bn(n_steps * 64 * 32, 0);
                std::vector<NarrowCol> ncols(P);
                std::vector<int> q(S);
                auto put_slot = [&](uint64_t slot, const std::vector<int>& v) {
                    for (uint64_t g = 0; g < n_kgroups; g++)
                        for (uint64_t jj = 0; jj < 4; jj++)
                            for (uint64_t k = 0; k < 128; k++) {
                                // FP4 side: k = 32 kb + 8 q + e' <-> bit 4 e' + jj of dword q of the lane's 16 bytes
                                const uint64_t kbA = k / 32, e = k % 32, smp = 512 * g + 128 * kbA + 32 * (e / 8) + 4 * (e % 8) + jj;
                                if (smp >= S) continue;
                                // FP8 side: lane kb = (k % 64) / 16, byte (k / 64) * 16 + k % 16
                                const uint64_t lane = slot + 16 * ((k % 64) / 16), byte = (k / 64) * 16 + k % 16;
                                Bn[((g * 4 + jj) * 64 + lane) * 32 + byte] = e4m3(v[smp]);
                            }
                };
                for (uint64_t j = 0; j < P; j++) {
                    const double Nd = (double)S, sum = (double)sums[j];
                    const double c = sum / Nd;
                    double mx = 0, A = 0;
                    std::vector<double> t(S);
                    for (uint64_t i = 0; i < S; i++) {
                        t[i] = (double)s->Y[j * S + i] - c;
                        mx = std::max(mx, std::fabs(t[i]));
                        A += std::fabs((double)s->Y[j * S + i]);
                    }
                    double u = mx > 0 ? mx / 15.0 : 1.0;
                    NarrowCol& nc = ncols[j];
                    for (int k = 0; k < NARROW_SLICES; k++) {
                        for (uint64_t i = 0; i < S; i++) {
                            int v = (int)std::lrint(t[i] / u);
                            v = std::max(-15, std::min(15, v));
                            q[i] = v;
                            t[i] -= u * v;
                        }
                        put_slot(4 * j + k, q);  // operand row 4 p + k; 4 p + 3 = ones
                        nc.w[k] = 2.0 * u;
                        u /= 30.0;
                    }
                    double rpos = 0, rneg = 0, rmax = 0;
                    for (uint64_t i = 0; i < S; i++) {
                        if (t[i] > 0) rpos += t[i]; else rneg -= t[i];
                        rmax = std::max(rmax, std::fabs(t[i]));
                    }
                    // (the slice products u_k * v and the running residual are evaluated in double: pad by their rounding)
                    const double fuzz = 64.0 * std::ldexp(1.0, -52) * (mx + std::fabs(c));
                    nc.t1 = Nd * c - sum;
                    nc.eg = (gamma * A * (1.0 + 1e-6) + 1e-12 * (1.0 + A)) * (1.0 + 1e-9);
                    nc.rall = (std::max(rpos, rneg) + Nd * fuzz) * (1.0 + 1e-9);
                    nc.rmax = (rmax + fuzz) * (1.0 + 1e-9);
                    // both sides evaluate N * x - N1 * sum and the slice sums in double: absolute slack of a few ulps of
                    // the largest intermediate (N * N * max|y|)
                    nc.pad = 256.0 * std::ldexp(1.0, -52) * Nd * Nd * (mx + std::fabs(c) + 1.0) + 1e-300;
                    // float32 pre-screen (score_narrow.hip): |r| <= N |ycf| (1 + 2^-10) + slackf: N1 |t1|, N E and the pad
                    // of the exact test, and the float32 roundings of the three products and sums.
                    {
                        const double slack = Nd * (nc.eg + std::min(nc.rall, Nd * nc.rmax)) + Nd * std::fabs(nc.t1) + nc.pad +
                                             Nd * std::ldexp(1.0, -20) * mx * Nd;
                        for (int k = 0; k < 3; k++) nc.wf[k] = (float)nc.w[k];
                        nc.slackf = std::nextafter((float)(slack * 1.001), std::numeric_limits<float>::infinity());
                    }
                }
                for (uint64_t j = 0; j < P; j++) put_slot(4 * j + 3, std::vector<int>(S, 1));
                s->d_Bn.alloc(Bn.size());
                s->d_ncols.alloc(P);
                KGWAS_HIP(hipMemcpy(s->d_Bn.p, Bn.data(), Bn.size(), hipMemcpyHostToDevice));
                KGWAS_HIP(hipMemcpy(s->d_ncols.p, ncols.data(), P * sizeof(NarrowCol), hipMemcpyHostToDevice));

            }
            // ---- block-scaled filter (score_mx.hip), the default: FP6 (+ FP4 / FP6) slices on the integer grids
            //   A6 = {0..15, 16..30 step 2, 32..60 step 4} (E2M3 x 8),  A4 = {0, 1, 2, 3, 4, 6, 8, 12} (E2M1 x 2):
            //   y_i - c ~ w * t_i,  t_i = a6_i (one slice), 8 a6_i + a4_i (FP4 second slice) or 32 a6_i + a6'_i (FP6 second
            //   slice); the accumulator is kappa * sum_i g_i t_i, kappa = 1/16, 1/4, 1/16, so one accumulator unit is
            //   u = w / kappa phenotype units and everything above (kalpha, the error terms in units of Dc) carries over
            //   with that u. The ones column has t = 1 / kappa: its accumulator is N1.
            // Which filter (KGWAS_COARSE_MX=1|0 forces one): the block-scaled one wherever its operands (1.25 bytes per
            // sample and column with two slices) leave a row no more LDS groups to pass through than the int8 filter's single
            // slice (1 byte) does. Measured: 1024 x 101 (one group each) 9.9 + 4.3 ms of filter + other kernels per 100 M
            // rows against 10.4 + 4.8; 2048 x 201 (five groups of three column tiles against four groups of four int8
            // tiles) 45.0 + 10.8 against 38.7 + 14.4 - there every row is loaded, expanded and tested once per group,
            // and the int8 filter keeps the shape.
            bool use_mx;
            if (const char* e = getenv("KGWAS_COARSE_MX")) {
                use_mx = atoi(e) != 0;
            } else {
                auto groups_for = [&](uint32_t tmax) {
                    uint64_t g = 1;
                    while (tmax && ((P + g - 1) / g + 1 + 15) / 16 > tmax) g++;
                    return tmax ? g : ~0ull;
                };
                const uint32_t steps = 4u * (uint32_t)(S / 512) + (uint32_t)((S % 512 + 127) / 128);
                uint32_t ctm = 0;
                for (uint32_t ct = 7; ct >= 1 && !ctm; ct--)
                    if (mx_lds_bytes(steps, ct, 2, 0) <= 160u * 1024u) ctm = ct;
                use_mx = groups_for(ctm) <= groups_for(s->coarse_T);
            }
            if (use_mx && !s->narrow && getenv("KGWAS_COARSE_SLICES") == nullptr) want[0] = false;  // one FP6 slice alone: only on request
            auto build_mx = [&](int mi) {
                const int ns = mi + 1;
                kgwas_scan::CoarseMode& M = s->cmode[mi];
                std::vector<int> G6, G4;  // the signed grids, ascending
                for (int q = 60; q >= 32; q -= 4) G6.push_back(-q);
                for (int q = 30; q >= 16; q -= 2) G6.push_back(-q);
                for (int q = 15; q >= -15; q--) G6.push_back(-q);
                for (int q = 16; q <= 30; q += 2) G6.push_back(q);
                for (int q = 32; q <= 60; q += 4) G6.push_back(q);
                for (int h : {-12, -8, -6, -4, -3, -2, -1, 0, 1, 2, 3, 4, 6, 8, 12}) G4.push_back(h);
                auto nearest = [](const std::vector<int>& g, double v) {  // index of the grid value closest to v
                    size_t hi = std::lower_bound(g.begin(), g.end(), v, [](int a, double b) { return (double)a < b; }) - g.begin();
                    if (hi == 0) return (size_t)0;
                    if (hi == g.size()) return g.size() - 1;
                    return (v - (double)g[hi - 1] <= (double)g[hi] - v) ? hi - 1 : hi;
                };
                auto e2m3 = [](int q) -> uint32_t {  // E2M3 code of q / 8
                    const uint32_t sg = q < 0 ? 0x20u : 0u;
                    const int a = std::abs(q);
                    if (a < 8) return sg | (uint32_t)a;
                    int e = 1, base = 8;
                    while (a >= 2 * base) base *= 2, e++;
                    return sg | ((uint32_t)e << 3) | (uint32_t)((a - base) / (base / 8));
                };
                auto e2m1 = [](int h) -> uint32_t {  // E2M1 code of h / 2
                    static const int tab[8] = {0, 1, 2, 3, 4, 6, 8, 12};
                    uint32_t i = 0;
                    while (tab[i] != std::abs(h)) i++;
                    return (h < 0 ? 8u : 0u) | i;
                };
                // whole 512-sample groups (the kernel reads all 64 bytes of those without a bounds check) + up to four quarter groups
                const uint32_t n_full = (uint32_t)(S / 512), nq = (uint32_t)((S % 512 + 127) / 128);
                const uint32_t n_steps = 4 * n_full + nq;
                // column tiles per LDS group the LDS can hold, and the LDS groups that takes, for a second-slice format
                auto ct_max = [&](uint32_t fp6) {
                    for (uint32_t ct = 7; ct >= 1; ct--)
                        if (mx_lds_bytes(n_steps, ct, (uint32_t)ns, fp6) <= 160u * 1024u) return ct;
                    return 0u;
                };
                auto groups_for = [&](uint32_t ctm) {
                    uint64_t g = 1;
                    while (((P + g - 1) / g + 1 + 15) / 16 > ctm) g++;
                    return g;
                };
                uint32_t s1_fp6 = 0;
                if (ns == 2 && ct_max(1) && ct_max(0) && groups_for(ct_max(1)) == groups_for(ct_max(0))) s1_fp6 = 1;
                if (const char* e = getenv("KGWAS_MX_S1")) s1_fp6 = atoi(e) == 6 && ct_max(1) ? 1u : 0u;  // experiments
                const uint32_t CTmax = ct_max(s1_fp6);
                if (!CTmax) throw Error(KGWAS_ERR_ARG, "coarse filter: too many accessions for the LDS");
                const int sh = ns == 1 ? 0 : (s1_fp6 ? 5 : 3);                 // t = 2^sh * a6 + a1
                const double kappa = (ns == 2 && !s1_fp6) ? 0.25 : 0.0625;     // accumulator = kappa * sum g t
                const int t_ones = (int)(1.0 / kappa);                         // in the LAST slice (a6 = 0 with two slices)
                const double t_max = ns == 1 ? 60.0 : (s1_fp6 ? 32.0 * 60.0 + 60.0 : 8.0 * 60.0 + 12.0);
                const std::vector<int>& G1 = s1_fp6 ? G6 : G4;
                std::vector<int> a0(S), a1(S);
                auto quantise_mx = [&](uint64_t j, CoarseCol& cc, ErrBound& eb) {
                    const double Nd = (double)S, sum = (double)sums[j];
                    const double c = sum / Nd;
                    double mx = 0, A = 0;
                    for (uint64_t i = 0; i < S; i++) {
                        const double y = (double)s->Y[j * S + i];
                        mx = std::max(mx, std::fabs(y - c));
                        A += std::fabs(y);
                    }
                    const double w = mx > 0 ? mx / t_max : 1.0;
                    const double u = w / kappa;  // one accumulator unit in phenotype units
                    double rpos = 0, rneg = 0, rmax = 0;
                    for (uint64_t i = 0; i < S; i++) {
                        const double y = (double)s->Y[j * S + i] - c;
                        const double x = y / w;
                        int b0 = 0, b1 = 0;
                        if (ns == 1) {
                            b0 = G6[nearest(G6, x)];
                        } else {
                            // the first slice's neighbours of x / 2^sh, each with its best second slice
                            const double sc = (double)(1 << sh);
                            const size_t k0 = nearest(G6, x / sc);
                            double best = 1e300;
                            for (size_t k = k0 ? k0 - 1 : 0; k <= std::min(k0 + 1, G6.size() - 1); k++) {
                                const int c1 = G1[nearest(G1, x - sc * G6[k])];
                                const double r = std::fabs(x - sc * G6[k] - c1);
                                if (r < best) best = r, b0 = G6[k], b1 = c1;
                            }
                        }
                        a0[i] = b0;
                        a1[i] = b1;
                        const double r = y - w * ((double)(1 << sh) * b0 + b1);
                        if (r > 0) rpos += r; else rneg -= r;
                        rmax = std::max(rmax, std::fabs(r));
                    }
                    const double rho = Nd * std::fabs(Nd * c - sum) * 2.0 + 1e-9 * (1.0 + std::fabs(sum));
                    const double Eg = gamma * A * (1.0 + 1e-6) + 1e-12 * (1.0 + A);
                    cc.kalpha = (1.0 - std::ldexp(1.0, -19)) / (Nd * u);
                    cc.iu = up(1.0 / u * (1.0 + 1e-6));
                    eb.eg = up((Eg + rho / Nd) * (1.0 + 1e-6) + 1e-30);
                    eb.rall = up(std::max(rpos, rneg) * (1.0 + 1e-6));
                    eb.rmax = up(rmax * (1.0 + 1e-6));
                    const double iu = 1.0 / u * (1.0 + 1e-6);
                    eb.egD = up((double)eb.eg * iu);
                    eb.rallD = up((double)eb.rall * iu);
                    eb.rmaxD = up((double)eb.rmax * iu);
                };
                // LDS groups: as few as hold all columns (+ a ones column each); or groups filled to the last slot and one
                // smaller launch for the rest when that multiplies fewer tiles
                uint64_t n_lgroups = groups_for(CTmax);
                uint64_t cper = (P + n_lgroups - 1) / n_lgroups;
                struct Plan {
                    uint64_t j0, n, CT, groups, cper;
                };
                std::vector<Plan> plan;
                plan.push_back(Plan{0, P, (cper + 1 + 15) / 16, n_lgroups, cper});
                if (n_lgroups > 1) {
                    const uint64_t cpf = (uint64_t)CTmax * 16 - 1;
                    const uint64_t full = P / cpf, rem = P - full * cpf;
                    const uint64_t CTr = rem ? (rem + 1 + 15) / 16 : 0;
                    if (full >= 1 && full + (rem ? 1 : 0) <= n_lgroups && full * CTmax + CTr < n_lgroups * plan[0].CT) {
                        plan.clear();
                        plan.push_back(Plan{0, full * cpf, CTmax, full, cpf});
                        if (rem) plan.push_back(Plan{full * cpf, rem, CTr, 1, rem});
                    }
                }
                M.mx = true;
                M.mx_full = n_full;
                M.mx_quarter = nq;
                M.mx_s1_fp6 = s1_fp6;
                M.mx_scale0 = 0x01010101u * (uint32_t)(0x7F + (ns == 1 ? 0 : 5));
                M.slices = (uint32_t)ns;
                M.n_parts = (uint32_t)plan.size();
                M.tile_slices = 0;
                uint32_t groups_all = 0;
                const uint32_t SB = mx_step_bytes_rt((uint32_t)ns, s1_fp6);
                for (size_t pi = 0; pi < plan.size(); pi++) {
                    const Plan& pl = plan[pi];
                    kgwas_scan::CoarsePart& Pt = M.part[pi];
                    const uint32_t CT = (uint32_t)pl.CT, slots = CT * 16;
                    Pt.T = CT;
                    Pt.n_lgroups = (uint32_t)pl.groups;
                    Pt.wide = false;
                    M.tile_slices += CT * (uint32_t)ns * (uint32_t)pl.groups;
                    groups_all += (uint32_t)pl.groups;
                    const size_t group_bytes = (size_t)n_steps * CT * SB;
                    std::vector<uint8_t> Bq(pl.groups * group_bytes, 0);
                    std::vector<CoarseCol> cols(pl.groups * slots);
                    for (auto& cc : cols) {
                        memset(&cc, 0, sizeof(cc));
                        cc.pheno = -1;
                    }
                    // the slice values of operand column `slot` of LDS group lg: v0 on the A6 grid, v1 on the second slice's
                    auto put = [&](uint64_t lg, uint64_t slot, const std::vector<int>& v0, const std::vector<int>& v1) {
                        const uint64_t t = slot / 16, n = slot % 16;
                        for (uint64_t st = 0; st < n_steps; st++) {
                            uint8_t* blk = &Bq[lg * group_bytes + (st * CT + t) * SB];
                            for (uint64_t kb = 0; kb < 4; kb++) {
                                const uint64_t lane = kb * 16 + n;
                                for (uint64_t e = 0; e < 32; e++) {
                                    // score_mx.hip: k = 32 kb + e <-> sample
                                    const uint64_t smp = st < 4ull * n_full ? 512 * (st / 4) + 128 * kb + 32 * (e / 8) + 4 * (e % 8) + st % 4
                                                                            : 512ull * n_full + 128 * (st - 4ull * n_full) + 32 * kb + 4 * (e % 8) + e / 8;
                                    if (smp >= S) continue;
                                    auto put6 = [&](uint8_t* part, int q) {  // 6-bit field e of the lane's 6 dwords: dwords 0-3 | 4-5
                                        const uint32_t code = e2m3(q);
                                        for (int b = 0; b < 6; b++)
                                            if (code & (1u << b)) {
                                                const uint64_t bit = 6 * e + b, dw = bit / 32;
                                                uint8_t* d = dw < 4 ? part + lane * 16 + dw * 4 : part + 1024 + lane * 8 + (dw - 4) * 4;
                                                d[(bit % 32) / 8] |= (uint8_t)(1u << (bit % 8));
                                            }
                                    };
                                    put6(blk, v0[smp]);
                                    if (ns == 2) {
                                        if (s1_fp6)
                                            put6(blk + 1536, v1[smp]);
                                        else
                                            blk[1536 + lane * 16 + e / 2] |= (uint8_t)(e2m1(v1[smp]) << (4 * (e % 2)));
                                    }
                                }
                            }
                        }
                    };
                    for (uint64_t j = pl.j0; j < pl.j0 + pl.n; j++) {
                        const uint64_t lg = (j - pl.j0) / pl.cper, slot = (j - pl.j0) % pl.cper;
                        CoarseCol& cc = cols[lg * slots + slot];
                        ErrBound eb;
                        quantise_mx(j, cc, eb);
                        M.eg_max = std::max(M.eg_max, eb.egD);
                        M.rall_max = std::max(M.rall_max, eb.rallD);
                        M.rmax_max = std::max(M.rmax_max, eb.rmaxD);
                        cc.pheno = (int32_t)j;
                        put(lg, slot, a0, a1);
                    }
                    {  // ones column: accumulator = N1
                        std::vector<int> ones(S, t_ones), zeros(S, 0);
                        for (uint64_t lg = 0; lg < pl.groups; lg++) put(lg, slots - 1, ns == 1 ? ones : zeros, ones);
                    }
                    Pt.d_Bq.alloc(Bq.size());
                    Pt.d_cols.alloc(cols.size());
                    KGWAS_HIP(hipMemcpy(Pt.d_Bq.p, Bq.data(), Bq.size(), hipMemcpyHostToDevice));
                    KGWAS_HIP(hipMemcpy(Pt.d_cols.p, cols.data(), cols.size() * sizeof(CoarseCol), hipMemcpyHostToDevice));
                }
                s->st.coarse_mode_tiles[mi] = M.part[0].T;
                s->st.coarse_mode_lgroups[mi] = groups_all;
                s->st.coarse_mode_tile_slices[mi] = M.tile_slices;
                s->st.coarse_mx = 1;
                s->st.coarse_mx_s1_fp6 = s1_fp6;
                s->st.coarse_mx_steps = n_steps;
                M.ready = true;
            };
            for (int mi = 0; mi < 2; mi++) {
                if (!want[mi]) continue;
                if (use_mx) {
                    build_mx(mi);
                    continue;
                }
                const int ns = mi + 1;
                kgwas_scan::CoarseMode& M = s->cmode[mi];
                // Operand columns ("slots") per LDS group: the group's share of the phenotype columns, padding, and
                // the ones column in the last slot (its dot product is the row's masked popcount N1).
                uint32_t Tmax = s->coarse_T;  // largest tile count whose operands fit the LDS
                if (ns == 2) Tmax &= ~1u;
                uint64_t n_lgroups = 1, cper = P;
                uint32_t T = 0;
                for (;; n_lgroups++) {
                    cper = (P + n_lgroups - 1) / n_lgroups;  // phenotype columns per group
                    T = (uint32_t)(ns * ((cper + 1 + 15) / 16));
                    if (T <= Tmax) break;
                }
                // plan[i] = {first column, columns, T, LDS groups, columns per group}
                struct Plan {
                    uint64_t j0, n, T, groups, cper;
                    bool wide;
                };
                std::vector<Plan> plan;
                plan.push_back(Plan{0, P, T, n_lgroups, cper, false});
                if (n_lgroups > 1) {
                    // The balanced split pads every group (201 columns, 4 tiles per group: 4 x (51 + ones) of 4 x 64
                    // slots = 16 tiles for 13 tiles' worth of columns). Alternative: groups filled to the last slot and
                    // ONE smaller launch for the rest - taken when it multiplies fewer tiles with no more row passes.
                    const uint64_t cpf = (uint64_t)(Tmax / (uint32_t)ns) * 16 - 1;  // columns of a full group
                    const uint64_t full = P / cpf, rem = P - full * cpf;
                    const uint64_t Tr = rem ? (uint64_t)ns * ((rem + 1 + 15) / 16) : 0;
                    static const bool no_split = getenv("KGWAS_COARSE_NOSPLIT") != nullptr;  // experiments
                    if (full >= 1 && full + (rem ? 1 : 0) <= n_lgroups && full * Tmax + Tr < n_lgroups * T && !no_split) {
                        plan.clear();
                        plan.push_back(Plan{0, full * cpf, Tmax, full, cpf, false});
                        if (rem) plan.push_back(Plan{full * cpf, rem, Tr, 1, rem, false});
                    }
                }
                // One slice, more tiles than the LDS holds at once, at most 14: the wide kernel keeps every tile's
                // accumulators in registers and streams the operands (score_wide.hip) - one group, every row expanded once.
                {
                    const uint32_t tiles = (uint32_t)((P + 1 + 15) / 16);
                    // Measured at 2048 samples x 201 columns (T = 13): 34.5 ms per 75.6 M rows against 32.0 ms for
                    // coarse_kernel's four LDS groups - the matrix pipe is busy 37 % of the time (one wave per SIMD: its
                    // epilogue, the stage barriers and the vector instructions beside the MFMAs are all exposed) - so
                    // it stays opt-in (KGWAS_WIDE=1) until it wins.
                    const bool on = getenv("KGWAS_WIDE") && atoi(getenv("KGWAS_WIDE")) == 1;
                    if (ns == 1 && n_lgroups > 1 && tiles >= 9 && tiles <= 14 && wide_lds_bytes(tiles) <= 160u * 1024u && on) {
                        plan.clear();
                        plan.push_back(Plan{0, P, tiles, 1, P, true});
                    }
                }
                M.slices = (uint32_t)ns;
                M.n_parts = (uint32_t)plan.size();
                M.tile_slices = 0;
                uint32_t groups_all = 0;
                for (size_t pi = 0; pi < plan.size(); pi++) {
                    const Plan& pl = plan[pi];
                    kgwas_scan::CoarsePart& Pt = M.part[pi];
                    const uint32_t Tp = (uint32_t)pl.T;
                    const uint32_t PG = Tp / (uint32_t)ns, slots = PG * 16;
                    Pt.T = Tp;
                    Pt.n_lgroups = (uint32_t)pl.groups;
                    Pt.wide = pl.wide;
                    M.tile_slices += Tp * (uint32_t)pl.groups;
                    groups_all += (uint32_t)pl.groups;
                    std::vector<int8_t> Bq(pl.groups * n_kgroups * 8ull * Tp * 1024ull, 0);
                    std::vector<CoarseCol> cols(pl.groups * slots);
                    for (auto& cc : cols) {
                        memset(&cc, 0, sizeof(cc));
                        cc.pheno = -1;
                    }
                    auto put = [&](uint64_t lg, uint64_t slot, const std::vector<int>& v0, const std::vector<int>& v1) {
                        const uint64_t pgl = slot / 16, n = slot % 16;
                        for (uint64_t g = 0; g < n_kgroups; g++)
                            for (uint64_t jj = 0; jj < 8; jj++)
                                for (uint64_t kg = 0; kg < 4; kg++)
                                    for (uint64_t e = 0; e < 16; e++) {
                                        // k-element e of step jj <-> sample (score_coarse.hip: expand_step)
                                        const uint64_t smp = 512 * g + 128 * kg + 32 * (e / 4) + 8 * (e % 4) + jj;
                                        if (smp >= S) continue;
                                        const uint64_t lane = kg * 16 + n;
                                        const uint64_t base = (((lg * n_kgroups + g) * 8 + jj) * Tp);
                                        if (ns == 1) {
                                            Bq[((base + pgl) * 64 + lane) * 16 + e] = (int8_t)v0[smp];
                                        } else {
                                            Bq[((base + 2 * pgl) * 64 + lane) * 16 + e] = (int8_t)v0[smp];
                                            Bq[((base + 2 * pgl + 1) * 64 + lane) * 16 + e] = (int8_t)v1[smp];
                                        }
                                    }
                    };
                    for (uint64_t j = pl.j0; j < pl.j0 + pl.n; j++) {
                        const uint64_t lg = (j - pl.j0) / pl.cper, slot = (j - pl.j0) % pl.cper;
                        CoarseCol& cc = cols[lg * slots + slot];
                        ErrBound eb;
                        quantise(j, ns, cc, eb);
                        M.eg_max = std::max(M.eg_max, eb.egD);
                        M.rall_max = std::max(M.rall_max, eb.rallD);
                        M.rmax_max = std::max(M.rmax_max, eb.rmaxD);
                        cc.pheno = (int32_t)j;
                        put(lg, slot, q0, q1);
                    }
                    {  // ones column: Dc = N1 (one slice: q0 = 1; two slices: Dc = 254*D0 + D1 with q0 = 0, q1 = 1)
                        std::vector<int> ones(S, 1), zeros(S, 0);
                        for (uint64_t lg = 0; lg < pl.groups; lg++) put(lg, slots - 1, ns == 1 ? ones : zeros, ones);
                    }
                    Pt.d_Bq.alloc(Bq.size());
                    Pt.d_cols.alloc(cols.size());
                    KGWAS_HIP(hipMemcpy(Pt.d_Bq.p, Bq.data(), Bq.size(), hipMemcpyHostToDevice));
                    KGWAS_HIP(hipMemcpy(Pt.d_cols.p, cols.data(), cols.size() * sizeof(CoarseCol), hipMemcpyHostToDevice));
                }
                s->st.coarse_mode_tiles[mi] = M.part[0].T;
                s->st.coarse_mode_lgroups[mi] = groups_all;
                s->st.coarse_mode_tile_slices[mi] = M.tile_slices;
                M.ready = true;
            }
            s->key_slots = (uint32_t)std::min<uint64_t>((uint64_t)s->cap * P, 0x7FFFFFFFull);
            s->d_surv_sorted.alloc(s->key_slots);
            s->bitmap_words = (s->chunk_max + 63) / 64;
            s->d_bitmap.alloc(P * s->bitmap_words);
            s->d_bm_blocks.alloc(P * ((s->bitmap_words + 1023) / 1024 + 1) + 4);
            s->d_surv_cnt.alloc(P);
            s->d_surv_off.alloc(P);
            s->d_key_count.alloc(1);
            s->d_tile_pref.alloc(P + 1);
            s->d_tile_cnt.alloc((size_t)s->key_slots / 256 + P + 2);
            s->d_tile_off.alloc((size_t)s->key_slots / 256 + P + 2);
            s->d_tmp_score.alloc(s->key_slots);
            // The record copies run as blit kernels (rocprofv3 shows __amd_rocclr_copyBuffer, not SDMA transfers), and at
            // normal priority they are only dispatched in the gaps of the compute stream: behind a 0.75 ms filter launch
            // of a one-column scan, a chunk's 1 MB of records reached the host 1.3-2.8 ms after its counts. A high-priority
            // queue gets them onto the chip between the running launch's workgroups. KGWAS_COPY_PRIO=0: the old behaviour.
            {
                int least = 0, greatest = 0;
                KGWAS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
                const bool hi = !(getenv("KGWAS_COPY_PRIO") && atoi(getenv("KGWAS_COPY_PRIO")) == 0);
                KGWAS_HIP(hipStreamCreateWithPriority(&s->copy_stream, hipStreamNonBlocking, hi ? greatest : least));
            }
            s->row_key_bits = 1;
            while (s->row_key_bits < 32 && (1ull << s->row_key_bits) < s->chunk_max) s->row_key_bits++;
        }
        if (!s->direct) s->d_sq.alloc(s->chunk_max * 2 * W_m);

        {
            if (s->coarse) {
                // device side: 20 B x key_slots of HBM per slot, up to 4 GiB in all; host side: the record ring
                const uint64_t slot_bytes = (uint64_t)s->key_slots * 20;
                s->n_slots = (int)std::min<uint64_t>(MAX_SLOTS, std::max<uint64_t>(4, (4ull << 30) / std::max<uint64_t>(slot_bytes, 1)));
                s->ring_size = (size_t)std::max<uint64_t>(std::min<uint64_t>(1ull << 30, (uint64_t)s->n_slots * slot_bytes), 2 * slot_bytes + 4096);
                // (tests: a ring barely larger than one chunk's worst case, so that it wraps and fills up)
                if (getenv("KGWAS_RING_BYTES"))
                    s->ring_size = (size_t)std::max<uint64_t>(strtoull(getenv("KGWAS_RING_BYTES"), nullptr, 10), slot_bytes + 4096);
                s->ring.alloc(s->ring_size);
            } else {
                const uint64_t slot_bytes = (uint64_t)s->cap * P * sizeof(Cand);
                s->n_slots = (int)std::min<uint64_t>(16, std::max<uint64_t>(4, (1ull << 30) / std::max<uint64_t>(slot_bytes, 1)));
            }
        }
        for (int si = 0; si < s->n_slots + (s->coarse ? 1 : 0); si++) {
            const bool is_redo = si == s->n_slots;
            Slot& sl = is_redo ? s->redo : s->slot[si];
            if (s->coarse && !is_redo) {
                sl.d_so_score.alloc(s->key_slots);
                sl.d_so_kmer.alloc(s->key_slots);
                sl.d_so_row.alloc(s->key_slots);
                sl.d_meta.alloc(2 * P + 2);
                sl.h_meta.alloc(2 * P + 2);
                memset(sl.h_meta.p, 0, (2 * P + 2) * sizeof(uint32_t));
                KGWAS_HIP(hipEventCreateWithFlags(&sl.ev_counts, hipEventBlockingSync));
            } else {
                sl.cand.alloc((uint64_t)s->cap * P);
                sl.d_cand = sl.cand.dev();
            }
            sl.d_cnt.alloc(P);
            sl.h_cnt.alloc(P);
            sl.d_tested.alloc(TESTED_SHARDS);
            sl.h_tested.alloc(TESTED_SHARDS);
            KGWAS_HIP(hipEventCreate(&sl.ev_sq0));
            KGWAS_HIP(hipEventCreate(&sl.ev_k0));
            KGWAS_HIP(hipEventCreate(&sl.ev_k1));
            // (blocking wait: the control thread sleeps instead of spinning beside the replay workers)
            KGWAS_HIP(hipEventCreateWithFlags(&sl.ev_done, hipEventBlockingSync));
            KGWAS_HIP(hipEventCreate(&sl.ev_mid));
        }
        s->d_dense.alloc(P * s->dense_rows);
        s->h_dense.alloc(P * s->dense_rows);
        s->d_n1.alloc(s->dense_rows);
        s->h_n1.alloc(s->dense_rows);
        s->d_kmer.alloc(s->dense_rows);
        s->h_kmer.alloc(s->dense_rows);
        s->d_tested_dense.alloc(TESTED_SHARDS);

        make_heaps(s.get());
        s->hist.resize(P);
        s->keys.resize(P);
        s->col_ms.assign(P, 0.0);
        s->trace = getenv("KGWAS_TRACE") != nullptr;
        unsigned nt = p->host_threads ? p->host_threads : usable_cpus();
        if (const char* e = getenv("KGWAS_HOST_THREADS"))
            if (atoi(e) > 0) nt = (unsigned)atoi(e);
        nt = (unsigned)std::min<uint64_t>(nt, P);
        s->pool.reset(new Pool(nt, pick_replay_cpus(nt, s->device)));
        s->st.replay_threads = nt;
        s->ingest.producer_cpus_ = p->host_threads ? p->host_threads : usable_cpus();
        // Column groups of the replay. Worker w owns the columns w, w + T, ... of the first floor(P / T) * T columns,
        // in groups of at most MAX_LOCKSTEP (a group's heaps take their replacements in lockstep, and stay in their
        // worker's cache from chunk to chunk); the P mod T columns left over float: each is a group of its own that
        // whichever worker is furthest ahead takes, which evens out what a static map cannot (101 columns on 16
        // workers is 5 x 7 + 11 x 6: the 7-column workers set the pace, 17 % above the mean).
        {
            const uint64_t T = nt, base = P / T;
            const uint64_t MKc = (uint64_t)BestHeap::MAX_LOCKSTEP;
            uint64_t per = base ? (base + ((base + MKc - 1) / MKc) - 1) / ((base + MKc - 1) / MKc) : 0;  // balanced split
            if (const char* e = getenv("KGWAS_REPLAY_GROUP"))
                if (atoi(e) > 0 && per) per = std::min<uint64_t>((uint64_t)atoi(e), MKc);
            for (uint64_t w = 0; w < T && base; w++) {
                std::vector<uint32_t> cur;
                for (uint64_t i = 0; i < base; i++) {
                    cur.push_back((uint32_t)(i * T + w));
                    if (cur.size() == per || i + 1 == base) {
                        s->grp_cols.push_back(cur);
                        s->grp_home.push_back((int)w);
                        cur.clear();
                    }
                }
            }
            for (uint64_t j = base * T; j < P; j++) {
                s->grp_cols.push_back(std::vector<uint32_t>(1, (uint32_t)j));
                s->grp_home.push_back(-1);
            }
            s->n_groups = s->grp_cols.size();
            s->gstate.reset(new kgwas_scan::GroupState[s->n_groups]);
            s->slot_left.reset(new std::atomic<uint32_t>[MAX_SLOTS]);
            for (int i = 0; i < MAX_SLOTS; i++) s->slot_left[i].store(0);
            kgwas_scan* raw = s.get();
            s->rp_fn = [raw](size_t w) { replay_worker(raw, w); };
        }
        s->st.kernel_used = s->narrow ? (uint32_t)KGWAS_KERNEL_NARROW : s->coarse ? (uint32_t)KGWAS_KERNEL_COARSE : kern;
        s->st.direct_mode = s->direct ? 1 : 0;
        *out = s.release();
    });
}

int kgwas_scan_feed_device(kgwas_scan* s, const void* d_rows, uint64_t n_rows, uint64_t first_row, void* hip_stream) {
    return guarded([&] {
        if (!s || (!d_rows && n_rows)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_device: null argument");
        if (s->finished) throw Error(KGWAS_ERR_STATE, "scan already finished");
        KGWAS_HIP(hipSetDevice(s->device));
        // order our stream after whatever the caller queued on theirs (e.g. the generator kernel)
        KGWAS_HIP(hipEventRecord(s->ev_user, (hipStream_t)hip_stream));
        KGWAS_HIP(hipStreamWaitEvent(s->stream, s->ev_user, 0));
        feed_device_impl(s, reinterpret_cast<const uint64_t*>(d_rows), n_rows, first_row);
    });
}

namespace {

// Chunked, double-buffered ingest (ingest.h): piece k+1 is produced and copied while piece k is scored and
// replayed; results do not depend on the piece size (rows are scored in order, thresholds only ever lag).
void ingest_run(kgwas_scan* s, uint64_t n_rows, uint64_t first_row, const Ingest::Fill& fill) {
    if (n_rows == 0) {
        feed_device_impl(s, nullptr, 0, first_row);
        return;
    }
    s->ingest.run(1 + s->W_f, n_rows, s->chunk_max, s->stream, fill,
                  [&](const uint64_t* d_rows, uint64_t row_off, uint64_t cnt) {
                      feed_device_impl(s, d_rows, cnt, first_row + row_off);  // returns with the stream idle
                  });
}

}  // namespace

int kgwas_scan_feed_host(kgwas_scan* s, const uint64_t* rows, uint64_t n_rows, uint64_t first_row) {
    return guarded([&] {
        if (!s || (!rows && n_rows)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_host: null argument");
        if (s->finished) throw Error(KGWAS_ERR_STATE, "scan already finished");
        KGWAS_HIP(hipSetDevice(s->device));
        const uint64_t stride = 1 + s->W_f;
        ingest_run(s, n_rows, first_row, [&](uint64_t* dst, uint64_t row_off, uint64_t cnt) {
            memcpy(dst, rows + row_off * stride, cnt * stride * 8);
        });
    });
}

int kgwas_scan_feed_table(kgwas_scan* s, kgwas_table* t, uint64_t row0, uint64_t n_rows) {
    return guarded([&] {
        if (!s || !t) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_table: null argument");
        if (s->finished) throw Error(KGWAS_ERR_STATE, "scan already finished");
        uint64_t n_acc = 0, t_rows = 0, wpr = 0;
        uint32_t k = 0;
        if (kgwas_table_info(t, &n_acc, &t_rows, &wpr, &k) != KGWAS_OK) throw Error(KGWAS_ERR_ARG, kgwas_last_error());
        if (n_acc != s->S_f) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_table: table and scan disagree on the accession count");
        if (row0 > t_rows || n_rows > t_rows - row0) throw Error(KGWAS_ERR_ARG, "kgwas_scan_feed_table: out of range");
        KGWAS_HIP(hipSetDevice(s->device));
        ingest_run(s, n_rows, row0, [&](uint64_t* dst, uint64_t row_off, uint64_t cnt) {
            if (kgwas_table_read_rows(t, row0 + row_off, cnt, dst) != KGWAS_OK) throw Error(KGWAS_ERR_IO, kgwas_last_error());
        });
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
        // A worker pops its columns (w, w+T, ...) up to eight at a time in lockstep where their sizes agree.
        const size_t Tw = s->pool->size();
        s->pool->parallel_for(std::min<size_t>(Tw, s->n_pheno), [&](size_t w) {
            std::vector<size_t> mine;
            for (size_t j = w; j < s->n_pheno; j += Tw) mine.push_back(j);
            size_t i = 0;
            while (i < mine.size()) {
                size_t K = 1;
                while (K < 8 && i + K < mine.size() && s->heaps[mine[i + K]].size() == s->heaps[mine[i]].size()) K++;
                const BestHeap* hp[8];
                std::vector<uint64_t>*km[8], *rw[8];
                std::vector<double>* sc[8];
                for (size_t k = 0; k < K; k++) {
                    const size_t j = mine[i + k];
                    hp[k] = &s->heaps[j];
                    km[k] = &s->res_kmer[j];
                    sc[k] = &s->res_score[j];
                    rw[k] = &s->res_row[j];
                }
                BestHeap::pop_all_n((int)K, hp, km, sc, rw);
                i += K;
            }
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
        refresh_full(s);
        // The device's thresholds and score histograms describe the rows behind the heaps that were just replaced:
        // start them again from the imported minima before anything else is fed (the next sparse chunk re-bases the
        // histograms, start_histograms).
        KGWAS_HIP(hipSetDevice(s->device));
        s->hist_ready = false;
        upload_thresholds(s);
    });
}

int kgwas_scan_lowest(const kgwas_scan* s, double* lowest, uint8_t* full) {
    return guarded([&] {
        if (!s || !lowest || !full) throw Error(KGWAS_ERR_ARG, "kgwas_scan_lowest: null argument");
        for (uint64_t j = 0; j < s->n_pheno; j++) {
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
