// scan_host.cpp — the host the session runs on: which CPUs the replay workers are pinned to, how many CPUs the process
// may really use (cgroup quota), the heaps' huge-page arena.
#include "scan_internal.h"
#include <sched.h>

namespace kgwas {

thread_local hipEvent_t g_bound_stop = nullptr;  // launch.h


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
    const char* e = opt_str("KGWAS_PIN_THREADS");
    if (e) mode = atoi(e);
    if (mode == 0) return none;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return none;
    // A process under a CPU-time quota smaller than its CPU set (a container on a shared host: 16 CPUs' worth of time on any of
    // 256) owns no core: the ones it would pin its workers to run other tenants' threads too, and a pinned worker that loses its
    // CPU for a few ms holds up its columns - or the whole dense fill - where an unpinned one is moved. Measured on test boxes
    // with a load average of 35-50 (round 6): headline step mean 18.7-19.5 ms pinned against 18.3-18.5 unpinned (medians 18.4
    // and 18.3: the pinned runs' outliers), the 16 MB heaps of -n 1000000 0.80 s against 0.50-0.71 per 100 M rows. The default
    // is therefore: one core each where the quota covers the CPU set (a node of one's own), none otherwise; KGWAS_PIN_THREADS=1..3
    // pins regardless, =0 never.
    if (!e && usable_cpus() < (unsigned)CPU_COUNT(&allowed)) {
        if (opt_str("KGWAS_TRACE")) fprintf(stderr, "[kgwas] replay workers not pinned: a quota of %u CPUs on a set of %d\n", usable_cpus(), CPU_COUNT(&allowed));
        return none;
    }
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
    if (opt_str("KGWAS_TRACE")) {
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
    {  // a taskset / cpuset narrower than the machine: pinned, spinning workers beyond it would only share its CPUs
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) n = std::min<unsigned>(n, (unsigned)c);
        }
    }
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

void check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error(KGWAS_ERR_DEVICE,
                    "no HIP device available: libkgwas has no CPU fallback (hipGetDeviceCount: " +
                        std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
    if (device < 0 || device >= n) throw Error(KGWAS_ERR_ARG, "device ordinal out of range");
}

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
    static const bool no_huge = exp_set("KGWAS_NO_HUGE_HEAPS");  // experiments
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
