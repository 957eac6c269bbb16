// heap_guard.cpp — the run-time guard of the heap emulation (kgwas_heap_selfcheck).
//
// The reference keeps its results in a std::priority_queue<tuple<u64 kmer, double score, size_t row>, vector<...>,
// cmp_second> (src/kmer_general.h:113-128, src/best_associations_heap.cpp:43-59); which of several equal scores
// survives and the order equal scores pop in are decided by libstdc++'s std::push_heap / std::pop_heap. csrc/heap.h does
// those element moves by hand (16-byte entries, lockstep, integer compares). The tests pin it against the literal type -
// but only on the boxes the tests run on. On a user's machine with another libstdc++ the emulation would keep ITS tie
// order while the reference, built there, would take the library's: so once per process, before the first session is
// created, a fixed stream of ties, NaNs, negative and infinite scores goes through both - BestHeap (single pushes and
// the lockstep form) and the literal std::priority_queue of THIS process's libstdc++ - and any difference (a heap minimum
// along the way, a pop sequence) makes kgwas_scan_create / kgwas_heap_new / kgwas_snps_* fail with KGWAS_ERR_STATE
// instead of producing results in an order the reference would not.
#include <math.h>

#include <atomic>
#include <functional>
#include <limits>
#include <mutex>
#include <queue>
#include <tuple>
#include <vector>

#include "common.h"
#include "heap.h"

namespace kgwas {
namespace {

// the reference's shapes (src/kmer_general.h:113-128)
typedef std::tuple<uint64_t, double, size_t> RefEntry;
struct RefCmp {
    bool operator()(const RefEntry& a, const RefEntry& b) const { return std::get<1>(a) > std::get<1>(b); }
};
struct RefCmpOther {  // test hook: a DIFFERENT tie rule - what a heap algorithm that moves equal elements differently looks like
    bool operator()(const RefEntry& a, const RefEntry& b) const { return std::get<1>(a) >= std::get<1>(b); }
};

template <class Cmp>
struct RefHeap {  // BestAssociationsHeap::add_association, literally (src/best_associations_heap.cpp:43-59)
    std::priority_queue<RefEntry, std::vector<RefEntry>, Cmp> q;
    size_t n_res;
    double lowest = 0;
    explicit RefHeap(size_t n) : n_res(n) {}
    void add(uint64_t k, double score, size_t row) {
        if (q.size() < n_res) {
            q.push(RefEntry(k, score, row));
            lowest = std::get<1>(q.top());
        } else if (score > lowest) {
            q.pop();
            q.push(RefEntry(k, score, row));
            lowest = std::get<1>(q.top());
        }
    }
};

struct Stream {
    std::vector<double> score;
    size_t topn;
};

// splitmix64: the stream must not depend on any library's generator
inline uint64_t mix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

std::vector<Stream> make_streams() {
    std::vector<Stream> out;
    uint64_t seed = 20240601;
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    // 1: seven distinct values only - every push decides a tie; 2: rising with repeats (every push replaces);
    // 3: NaN, -0.0, negative and +inf among ties (the float compare path); 4: a heap that never fills; 5: size 1
    for (int kind = 0; kind < 5; kind++) {
        Stream s;
        s.topn = kind == 3 ? 4096 : kind == 4 ? 1 : (size_t)(97 + 80 * kind);
        for (int i = 0; i < 2000; i++) {
            const uint64_t r = mix(seed);
            double v;
            switch (kind) {
                case 0: v = (double)(r % 7) * 0.25; break;
                case 1: v = (double)(i / 3) + (double)(r % 2) * 0.5; break;
                case 2: {
                    const unsigned c = (unsigned)(r % 16);
                    v = c == 0 ? nan : c == 1 ? inf : c == 2 ? -0.0 : c == 3 ? -1.5 : c == 4 ? 0.0 : (double)((r >> 8) % 9);
                    break;
                }
                default: v = (double)(r % 5); break;
            }
            s.score.push_back(v);
        }
        out.push_back(std::move(s));
    }
    return out;
}

inline bool same_bits(double a, double b) {
    uint64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    return x == y;
}

template <class Cmp>
bool run_check(std::string& why) {
    const std::vector<Stream> streams = make_streams();
    std::vector<uint64_t> k, r;
    std::vector<double> sc;
    for (size_t si = 0; si < streams.size(); si++) {
        const Stream& st = streams[si];
        BestHeap mine(st.topn);
        RefHeap<Cmp> ref(st.topn);
        for (size_t i = 0; i < st.score.size(); i++) {
            mine.add(1000 + i, st.score[i], i);
            ref.add(1000 + i, st.score[i], i);
            if (mine.size() != ref.q.size() || !same_bits(mine.lowest(), ref.lowest)) {
                why = "stream " + std::to_string(si) + ": heap minimum differs after push " + std::to_string(i);
                return false;
            }
        }
        mine.pop_all(k, sc, r);
        for (size_t i = 0; i < k.size(); i++) {
            const RefEntry e = ref.q.top();
            ref.q.pop();
            if (std::get<0>(e) != k[i] || !same_bits(std::get<1>(e), sc[i]) || std::get<2>(e) != r[i]) {
                why = "stream " + std::to_string(si) + ": pop " + std::to_string(i) + " differs";
                return false;
            }
        }
        if (!ref.q.empty()) {
            why = "stream " + std::to_string(si) + ": sizes differ";
            return false;
        }
    }
    // the lockstep form: three full heaps of equal size take their replacements together (scan_replay.cpp)
    {
        const size_t n = 129;
        BestHeap h0(n), h1(n), h2(n);
        BestHeap* hp[3] = {&h0, &h1, &h2};
        RefHeap<Cmp> q0(n), q1(n), q2(n);
        RefHeap<Cmp>* qp[3] = {&q0, &q1, &q2};
        uint64_t seed = 7;
        for (size_t i = 0; i < 1500; i++) {
            uint64_t km[3], row[3];
            double s[3];
            bool all_beat = true;
            for (int j = 0; j < 3; j++) {
                km[j] = 5 * i + j;
                row[j] = i;
                s[j] = (double)((mix(seed) % 11) + i / 200);
                all_beat = all_beat && hp[j]->full() && s[j] > hp[j]->lowest();
            }
            if (all_beat)
                BestHeap::replace_top_n(3, hp, km, s, row);
            else
                for (int j = 0; j < 3; j++) hp[j]->add(km[j], s[j], row[j]);
            for (int j = 0; j < 3; j++) qp[j]->add(km[j], s[j], row[j]);
        }
        for (int j = 0; j < 3; j++) {
            hp[j]->pop_all(k, sc, r);
            for (size_t i = 0; i < k.size(); i++) {
                const RefEntry e = qp[j]->q.top();
                qp[j]->q.pop();
                if (std::get<0>(e) != k[i] || !same_bits(std::get<1>(e), sc[i]) || std::get<2>(e) != r[i]) {
                    why = "lockstep heap " + std::to_string(j) + ": pop " + std::to_string(i) + " differs";
                    return false;
                }
            }
        }
    }
    return true;
}

std::once_flag g_once;
bool g_ok = false;
std::string g_why;

}  // namespace

// Throws KGWAS_ERR_STATE if the emulation and this process's std::priority_queue disagree (checked once, ~0.3 ms).
void require_heap_emulation() {
    std::call_once(g_once, [] { g_ok = run_check<RefCmp>(g_why); });
    if (!g_ok)
        throw Error(KGWAS_ERR_STATE, "the heap emulation (csrc/heap.h) does not reproduce this libstdc++'s std::priority_queue: " + g_why +
                                         " - tie order would differ from the reference's; refusing to run");
}

}  // namespace kgwas

extern "C" int kgwas_heap_selfcheck(uint32_t flags) {
    using namespace kgwas;
    return guarded([&] {
        if (flags & 1u) {  // test hook: hold the emulation against a reference with a different tie rule - must be noticed
            std::string why;
            if (!run_check<RefCmpOther>(why)) throw Error(KGWAS_ERR_STATE, "heap self-check (altered reference): " + why);
            return;
        }
        require_heap_emulation();
    });
}
