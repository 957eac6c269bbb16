// heap_one_bench.cpp - ONE heap's replace-min through heap.h's BestHeap::add (the one-column scan's replay: a single thread,
// 101 k effective pushes per 100 M rows), against a literal std::priority_queue: time per push and identical pop order.
// build: g++ -O3 -march=native -std=c++17 -I kmersgwas_amd/csrc tools/heap_one_bench.cpp -o /tmp/heap_one_bench
#include <chrono>
#include <cstdio>
#include <queue>
#include <random>
#include <vector>

#include "heap.h"

struct QE {
    double score;
    uint64_t kmer, row;
};
struct QG {
    bool operator()(const QE& l, const QE& r) const { return l.score > r.score; }
};

int main(int argc, char** argv) {
    const size_t N = argc > 1 ? atoi(argv[1]) : 10001;
    const size_t pushes = argc > 2 ? atoi(argv[2]) : 2000000;
    const int ties = argc > 3 ? atoi(argv[3]) : 0;  // 1: scores on a coarse grid (ties everywhere)
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    // the stream: every score above the current minimum (as the replay sees after the device's filter)
    std::vector<double> sc(N + pushes);
    for (size_t i = 0; i < N; i++) sc[i] = ties ? (double)(rng() % 512) : U(rng);
    kgwas::BestHeap h(N);
    std::priority_queue<QE, std::vector<QE>, QG> q;
    for (size_t i = 0; i < N; i++) {
        h.add(i, sc[i], i);
        q.push(QE{sc[i], i, i});
    }
    double low = h.lowest();
    for (size_t i = N; i < N + pushes; i++) {
        const double hi = ties ? 4096.0 : 1.0;
        double x = low + (hi - low) * U(rng);
        if (ties) x = (double)(long)x + 1.0;
        if (!(x > low)) x = low + 1e-9;
        sc[i] = x;
        // the reference's rule on the mirror
        if (x > q.top().score) {
            q.pop();
            q.push(QE{x, i, i});
        }
        low = q.top().score;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = N; i < N + pushes; i++) h.add(i, sc[i], i);
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (double)pushes;
    std::vector<uint64_t> km, rw;
    std::vector<double> ss;
    h.pop_all(km, ss, rw);
    std::vector<uint64_t> mk;
    while (!q.empty()) {
        mk.push_back(q.top().kmer);
        q.pop();
    }
    bool fwd = km.size() == mk.size(), rev = fwd;
    for (size_t i = 0; i < km.size() && (fwd || rev); i++) {
        fwd = fwd && km[i] == mk[i];
        rev = rev && km[i] == mk[mk.size() - 1 - i];
    }
    const bool same = fwd || rev;
    printf("N %zu, %zu pushes%s: %.1f ns per push (effective %llu), pop order %s\n", N, pushes, ties ? " (tie-heavy)" : "", ns,
           (unsigned long long)h.pushes(), same ? "as std::priority_queue" : "DIFFERENT");
    return same ? 0 : 1;
}
