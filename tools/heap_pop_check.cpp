// heap_pop_check.cpp - BestHeap::pop_all on heaps with planted equal scores against a literal std::priority_queue: the pops for real up to the
// last tied score, sorted from there (heap.h). build: g++ -O2 -std=c++17 -I kmersgwas_amd/csrc tools/heap_pop_check.cpp -o /tmp/heap_pop_check
#include <cstdio>
#include <queue>
#include <random>
#include <vector>
#include "heap.h"
struct QE { double score; uint64_t id; };
struct QG { bool operator()(const QE& l, const QE& r) const { return l.score > r.score; } };
int main() {
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    int bad = 0, hybrid_cases = 0;
    for (int trial = 0; trial < 400; trial++) {
        const size_t N = 50 + rng() % 3000, n = N + rng() % (6 * N);
        std::vector<double> sc(n);
        for (auto& x : sc) x = 10.0 + U(rng);
        const int pairs = 1 + rng() % 4;
        for (int k = 0; k < pairs; k++) sc[rng() % n] = sc[rng() % n];  // plant equal scores anywhere
        kgwas::BestHeap h(N);
        std::priority_queue<QE, std::vector<QE>, QG> q;
        for (size_t i = 0; i < n; i++) {
            h.add(i, sc[i], i);
            if (q.size() < N) q.push(QE{sc[i], i});
            else if (sc[i] > q.top().score) { q.pop(); q.push(QE{sc[i], i}); }
        }
        std::vector<uint64_t> km, rw; std::vector<double> ss;
        h.pop_all(km, ss, rw);
        std::vector<uint64_t> mk; std::vector<double> ms;
        while (!q.empty()) { mk.push_back(q.top().id); ms.push_back(q.top().score); q.pop(); }
        bool same = km.size() == mk.size();
        for (size_t i = 0; same && i < km.size(); i++) same = km[i] == mk[i] && ss[i] == ms[i] && rw[i] == mk[i];
        if (!same) bad++;
        // was there a tie among the kept entries below the top eighth?
        for (size_t i = 1; i < ms.size(); i++) if (ms[i] == ms[i - 1] && i + 1 <= ms.size() - ms.size() / 8) { hybrid_cases++; break; }
    }
    printf("400 random heaps with planted equal scores: %d differ from std::priority_queue; %d had a tie below the top eighth (hybrid pops)\n", bad, hybrid_cases);
    return bad ? 1 : 0;
}
