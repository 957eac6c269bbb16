// heap_soa_bench.cpp — microbenchmark (not part of the product): the replay's K-way lockstep heap update on 16-byte
// (score, slot) entries (heap.h today) against a split layout, scores f64 and slots u32 in two arrays (half the bytes
// on the compare path). T threads x H = K heaps of N entries; checks that both end in the same layout.
// build: g++ -O3 -march=native -std=c++17 -pthread tools/heap_soa_bench.cpp -o tools/bin/heap_soa_bench
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <random>
#include <thread>
#include <vector>
#include <sys/mman.h>

struct Ent {
    double score;
    uint32_t slot;
};

template <int K>
static inline void replace_aos(Ent* const* a, ptrdiff_t n, Ent* x) {
    Ent v[K];
    ptrdiff_t h[K], c[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = a[k][n - 1];
        x[k].slot = a[k][0].slot;
        h[k] = 0;
        c[k] = 0;
    }
    for (;;) {
        bool any = false;
    #pragma unroll
    for (int k = 0; k < K; k++) {
            if (c[k] < lim) {
                ptrdiff_t cc = 2 * (c[k] + 1);
                cc -= (a[k][cc].score > a[k][cc - 1].score) ? 1 : 0;
                a[k][h[k]] = a[k][cc];
                h[k] = cc;
                c[k] = cc;
                any = true;
            }
        }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
            c[k] = 2 * (c[k] + 1);
            a[k][h[k]] = a[k][c[k] - 1];
            h[k] = c[k] - 1;
        }
        ptrdiff_t hh = h[k], p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].score > v[k].score) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = v[k];
        hh = n - 1;
        p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].score > x[k].score) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = x[k];
    }
}

// the same moves with the scores compared as int64 bit patterns (valid for non-negative, non-NaN doubles: IEEE order
// is integer order there): integer loads and compares have 4 + 1 cycles of latency where ucomisd from memory has 7 + 3
struct EntI {
    int64_t key;
    uint32_t slot;
};
template <int K>
static inline void replace_aos_int(EntI* const* a, ptrdiff_t n, EntI* x) {
    EntI v[K];
    ptrdiff_t h[K], c[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = a[k][n - 1];
        x[k].slot = a[k][0].slot;
        h[k] = 0;
        c[k] = 0;
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (c[k] < lim) {
                ptrdiff_t cc = 2 * (c[k] + 1);
                cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
                a[k][h[k]] = a[k][cc];
                h[k] = cc;
                c[k] = cc;
                any = true;
            }
        }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
            c[k] = 2 * (c[k] + 1);
            a[k][h[k]] = a[k][c[k] - 1];
            h[k] = c[k] - 1;
        }
        ptrdiff_t hh = h[k], p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > v[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = v[k];
        hh = n - 1;
        p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > x[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = x[k];
    }
}

// integer compares + both climbs in lockstep, branch-free per heap (one loop-exit misprediction per round, not per heap)
template <int K>
static inline void replace_aos_int2(EntI* const* a, ptrdiff_t n, EntI* x) {
    EntI v[K];
    ptrdiff_t h[K], c[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = a[k][n - 1];
        x[k].slot = a[k][0].slot;
        h[k] = 0;
        c[k] = 0;
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (c[k] < lim) {
                ptrdiff_t cc = 2 * (c[k] + 1);
                cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
                a[k][h[k]] = a[k][cc];
                h[k] = cc;
                c[k] = cc;
                any = true;
            }
        }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
            c[k] = 2 * (c[k] + 1);
            a[k][h[k]] = a[k][c[k] - 1];
            h[k] = c[k] - 1;
        }
    }
    for (int phase = 0; phase < 2; phase++) {
        ptrdiff_t hh[K];
        EntI val[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            hh[k] = phase == 0 ? h[k] : n - 1;
            val[k] = phase == 0 ? v[k] : x[k];
        }
        for (;;) {
            bool any = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const ptrdiff_t p = hh[k] > 0 ? (hh[k] - 1) / 2 : 0;
                const bool go = (hh[k] > 0) & (a[k][p].key > val[k].key);
                const EntI src = go ? a[k][p] : val[k];
                a[k][hh[k]] = src;  // (the final position gets the value itself, again and again once the heap is done)
                hh[k] = go ? p : hh[k];
                any |= go;
            }
            if (!any) break;
        }
    }
}

// integer compares; the hole walk without its per-level bounds test: every path has at least `sure` levels (all indices
// of depth d are below lim while 2^(d+1) - 2 < lim), only the last one or two are conditional; hole = the previous child
template <int K>
static inline void replace_aos_int3(EntI* const* a, ptrdiff_t n, EntI* x) {
    EntI v[K];
    ptrdiff_t c[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
    int sure = 0;
    while ((((ptrdiff_t)2) << sure) - 2 < lim) sure++;  // depths 0 .. sure - 1 are below lim whatever the path
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = a[k][n - 1];
        x[k].slot = a[k][0].slot;
        c[k] = 0;
    }
    for (int lvl = 0; lvl < sure; lvl++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            ptrdiff_t cc = 2 * (c[k] + 1);
            cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
            a[k][c[k]] = a[k][cc];
            c[k] = cc;
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        while (c[k] < lim) {
            ptrdiff_t cc = 2 * (c[k] + 1);
            cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
            a[k][c[k]] = a[k][cc];
            c[k] = cc;
        }
        ptrdiff_t hh = c[k];
        if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
            const ptrdiff_t c2 = 2 * (c[k] + 1);
            a[k][hh] = a[k][c2 - 1];
            hh = c2 - 1;
        }
        ptrdiff_t p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > v[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = v[k];
        hh = n - 1;
        p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > x[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = x[k];
    }
}

// integer compares, top-down: the hole stops where the old last leaf v belongs (first path node with key > v) instead of
// walking to a leaf and climbing back - the same final array (the path's keys never decrease downwards, so "climb while
// parent > v" ends exactly there)
template <int K>
static inline void replace_aos_int4(EntI* const* a, ptrdiff_t n, EntI* x) {
    EntI v[K];
    ptrdiff_t c[K];
    bool act[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = a[k][n - 1];
        x[k].slot = a[k][0].slot;
        c[k] = 0;
        act[k] = true;
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (act[k] && c[k] < lim) {
                ptrdiff_t cc = 2 * (c[k] + 1);
                cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
                if (a[k][cc].key > v[k].key) {
                    act[k] = false;  // v belongs at c[k]
                } else {
                    a[k][c[k]] = a[k][cc];
                    c[k] = cc;
                    any = true;
                }
            }
        }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        ptrdiff_t hh = c[k];
        if (act[k] && (len & 1) == 0 && c[k] == (len - 2) / 2) {  // the last inner node has a single (left) child
            const ptrdiff_t c2 = 2 * (c[k] + 1);
            if (!(a[k][c2 - 1].key > v[k].key)) {
                a[k][hh] = a[k][c2 - 1];
                hh = c2 - 1;
            }
        }
        a[k][hh] = v[k];
        hh = n - 1;
        ptrdiff_t p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > x[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = x[k];
    }
}

// integer compares; `sure` unconditional levels, then the remaining one or two levels BRANCH-FREE (a path ends at depth
// 12 or 13 with even odds at n = 10001: the bounds test of the last level is a coin flip for the predictor)
template <int K>
static inline void replace_aos_int5(EntI* const* a, ptrdiff_t n, EntI* x) {
    EntI v[K];
    ptrdiff_t c[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
    int sure = 0;
    while ((((ptrdiff_t)2) << sure) - 2 < lim) sure++;
    int extra = 0;  // levels that only some paths have
    while ((((ptrdiff_t)1) << (sure + extra)) - 1 < lim) extra++;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = a[k][n - 1];
        x[k].slot = a[k][0].slot;
        c[k] = 0;
    }
    for (int lvl = 0; lvl < sure; lvl++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            ptrdiff_t cc = 2 * (c[k] + 1);
            cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
            a[k][c[k]] = a[k][cc];
            c[k] = cc;
        }
    }
    for (int e = 0; e < extra; e++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool go = c[k] < lim;
            const ptrdiff_t base = go ? c[k] : 0;  // (a harmless in-range stand-in when this path has ended)
            ptrdiff_t cc = 2 * (base + 1);
            cc -= (a[k][cc].key > a[k][cc - 1].key) ? 1 : 0;
            const ptrdiff_t src = go ? cc : c[k];
            a[k][c[k]] = a[k][src];
            c[k] = src;
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        ptrdiff_t hh = c[k];
        if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
            const ptrdiff_t c2 = 2 * (c[k] + 1);
            a[k][hh] = a[k][c2 - 1];
            hh = c2 - 1;
        }
        ptrdiff_t p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > v[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = v[k];
        hh = n - 1;
        p = (hh - 1) / 2;
        while (hh > 0 && a[k][p].key > x[k].key) {
            a[k][hh] = a[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        a[k][hh] = x[k];
    }
}

template <int K, typename SlotT>
static inline void replace_soa(double* const* sc, SlotT* const* sl, ptrdiff_t n, const double* xs, SlotT* evicted) {
    double vs[K];
    SlotT vl[K], xl[K];
    ptrdiff_t h[K], c[K];
    const ptrdiff_t len = n - 1, lim = (len - 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        vs[k] = sc[k][n - 1];
        vl[k] = sl[k][n - 1];
        xl[k] = sl[k][0];
        evicted[k] = xl[k];
        h[k] = 0;
        c[k] = 0;
    }
    for (;;) {
        bool any = false;
    #pragma unroll
    for (int k = 0; k < K; k++) {
            if (c[k] < lim) {
                ptrdiff_t cc = 2 * (c[k] + 1);
                const double r = sc[k][cc], l = sc[k][cc - 1];
                const bool left = r > l;
                cc -= left ? 1 : 0;
                sc[k][h[k]] = left ? l : r;
                sl[k][h[k]] = sl[k][cc];
                h[k] = cc;
                c[k] = cc;
                any = true;
            }
        }
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if ((len & 1) == 0 && c[k] == (len - 2) / 2) {
            c[k] = 2 * (c[k] + 1);
            sc[k][h[k]] = sc[k][c[k] - 1];
            sl[k][h[k]] = sl[k][c[k] - 1];
            h[k] = c[k] - 1;
        }
        ptrdiff_t hh = h[k], p = (hh - 1) / 2;
        while (hh > 0 && sc[k][p] > vs[k]) {
            sc[k][hh] = sc[k][p];
            sl[k][hh] = sl[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        sc[k][hh] = vs[k];
        sl[k][hh] = vl[k];
        hh = n - 1;
        p = (hh - 1) / 2;
        while (hh > 0 && sc[k][p] > xs[k]) {
            sc[k][hh] = sc[k][p];
            sl[k][hh] = sl[k][p];
            hh = p;
            p = (hh - 1) / 2;
        }
        sc[k][hh] = xs[k];
        sl[k][hh] = xl[k];
    }
}

template <int K>
static void run(int T, int N, int pushes) {
    for (int variant = 0; variant < 10; variant++) {
        std::vector<double> ns(T);
        std::vector<uint64_t> chk(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                std::mt19937_64 rng(1234 + t);
                std::uniform_real_distribution<double> U(0.0, 1.0);
                std::vector<std::vector<Ent>> heaps(K);
                std::vector<std::vector<double>> hs(K);
                std::vector<std::vector<uint32_t>> hl32(K);
                std::vector<std::vector<uint16_t>> hl16(K);
                std::vector<std::vector<uint64_t>> km(K), rw(K);
                struct Greater {
                    bool operator()(const Ent& l, const Ent& r) const { return l.score > r.score; }
                };
                for (int h = 0; h < K; h++) {
                    for (int i = 0; i < N; i++) {
                        heaps[h].push_back(Ent{U(rng), (uint32_t)i});
                        std::push_heap(heaps[h].begin(), heaps[h].end(), Greater());
                    }
                    for (int i = 0; i < N; i++) {
                        hs[h].push_back(heaps[h][i].score);
                        hl32[h].push_back(heaps[h][i].slot);
                        hl16[h].push_back((uint16_t)heaps[h][i].slot);
                    }
                    km[h].resize(N);
                    rw[h].resize(N);
                }
                std::vector<double> u((size_t)pushes * K);
                for (auto& x : u) x = U(rng);
                auto t0 = std::chrono::steady_clock::now();
                if (variant == 0 || variant == 3) {
                    Ent* a[K];
                    uint64_t* kmp[K];
                    uint64_t* rwp[K];
                    for (int k = 0; k < K; k++) {
                        a[k] = heaps[k].data();
                        kmp[k] = km[k].data();
                        rwp[k] = rw[k].data();
                    }
                    if (variant == 3) {  // the same 16-byte entries (and the payload arrays) in one huge-page arena
                        const size_t bytes = ((size_t)K * N * (sizeof(Ent) + 16) + (4u << 20)) & ~((size_t)(2u << 20) - 1);
                        char* q = (char*)aligned_alloc(2u << 20, bytes);
                        madvise(q, bytes, MADV_HUGEPAGE);
                        memset(q, 0, bytes);
                        for (int k = 0; k < K; k++) {
                            memcpy(q, heaps[k].data(), (size_t)N * sizeof(Ent));
                            a[k] = (Ent*)q;
                            q += (size_t)N * sizeof(Ent);
                        }
                        for (int k = 0; k < K; k++) {
                            kmp[k] = (uint64_t*)q;
                            q += (size_t)N * 8;
                            rwp[k] = (uint64_t*)q;
                            q += (size_t)N * 8;
                        }
                        t0 = std::chrono::steady_clock::now();
                    }
                    for (int i = 0; i < pushes; i++) {
                        Ent x[K];
                        uint32_t s0[K];
                        for (int k = 0; k < K; k++) {
                            const double lo = a[k][0].score;
                            x[k] = Ent{lo + (1.0 - lo) * u[(size_t)k * pushes + i], 0};
                            s0[k] = a[k][0].slot;
                        }
                        replace_aos<K>(a, N, x);
                        for (int k = 0; k < K; k++) {
                            kmp[k][s0[k]] = i;
                            rwp[k][s0[k]] = i;
                        }
                    }
                    if (variant == 3)
                        for (int k = 0; k < K; k++) memcpy(heaps[k].data(), a[k], (size_t)N * sizeof(Ent));
                } else if (variant == 9) {  // integer compares; payloads appended to one sequential log instead of random slot writes
                    EntI* a[K];
                    for (int k = 0; k < K; k++) a[k] = reinterpret_cast<EntI*>(heaps[k].data());
                    std::vector<uint64_t> log((size_t)pushes * K * 2 + 16);
                    uint64_t* lp = log.data();
                    t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < pushes; i++) {
                        EntI x[K];
                        for (int k = 0; k < K; k++) {
                            double lo;
                            memcpy(&lo, &a[k][0].key, 8);
                            const double xv = lo + (1.0 - lo) * u[(size_t)k * pushes + i];
                            memcpy(&x[k].key, &xv, 8);
                            x[k].slot = 0;
                            lp[0] = ((uint64_t)a[k][0].slot << 32) | (uint32_t)i;
                            lp[1] = (uint64_t)i;
                            lp += 2;
                        }
                        replace_aos_int<K>(a, N, x);
                    }
                } else if (variant >= 4) {
                    EntI* a[K];
                    for (int k = 0; k < K; k++) a[k] = reinterpret_cast<EntI*>(heaps[k].data());
                    for (int i = 0; i < pushes; i++) {
                        EntI x[K];
                        uint32_t s0[K];
                        for (int k = 0; k < K; k++) {
                            double lo;
                            memcpy(&lo, &a[k][0].key, 8);
                            const double xv = lo + (1.0 - lo) * u[(size_t)k * pushes + i];
                            memcpy(&x[k].key, &xv, 8);
                            x[k].slot = 0;
                            s0[k] = a[k][0].slot;
                        }
                        if (variant == 4) replace_aos_int<K>(a, N, x); else if (variant == 5) replace_aos_int2<K>(a, N, x); else if (variant == 6) replace_aos_int3<K>(a, N, x); else if (variant == 7) replace_aos_int4<K>(a, N, x); else replace_aos_int5<K>(a, N, x);
                        for (int k = 0; k < K; k++) {
                            km[k][s0[k]] = i;
                            rw[k][s0[k]] = i;
                        }
                    }
                } else if (variant == 1) {
                    double* sc[K];
                    uint32_t* sl[K];
                #pragma unroll
    for (int k = 0; k < K; k++) {
                        sc[k] = hs[k].data();
                        sl[k] = hl32[k].data();
                    }
                    for (int i = 0; i < pushes; i++) {
                        double xs[K];
                        uint32_t ev[K];
                    #pragma unroll
    for (int k = 0; k < K; k++) {
                            const double lo = sc[k][0];
                            xs[k] = lo + (1.0 - lo) * u[(size_t)k * pushes + i];
                        }
                        replace_soa<K, uint32_t>(sc, sl, N, xs, ev);
                    #pragma unroll
    for (int k = 0; k < K; k++) {
                            km[k][ev[k]] = i;
                            rw[k][ev[k]] = i;
                        }
                    }
                } else {
                    double* sc[K];
                    uint16_t* sl[K];
                #pragma unroll
    for (int k = 0; k < K; k++) {
                        sc[k] = hs[k].data();
                        sl[k] = hl16[k].data();
                    }
                    for (int i = 0; i < pushes; i++) {
                        double xs[K];
                        uint16_t ev[K];
                    #pragma unroll
    for (int k = 0; k < K; k++) {
                            const double lo = sc[k][0];
                            xs[k] = lo + (1.0 - lo) * u[(size_t)k * pushes + i];
                        }
                        replace_soa<K, uint16_t>(sc, sl, N, xs, ev);
                    #pragma unroll
    for (int k = 0; k < K; k++) {
                            km[k][ev[k]] = i;
                            rw[k][ev[k]] = i;
                        }
                    }
                }
                const double el = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
                ns[t] = el / ((double)pushes * K);
                uint64_t c = 0;
                for (int k = 0; k < K; k++)
                    for (int i = 0; i < N; i++) {
                        uint64_t sbits;
                        const double s = (variant == 0 || variant >= 3) ? heaps[k][i].score : hs[k][i];
                        const uint32_t slot = (variant == 0 || variant >= 3) ? heaps[k][i].slot : variant == 1 ? hl32[k][i] : hl16[k][i];
                        memcpy(&sbits, &s, 8);
                        c = c * 1099511628211ull + (sbits ^ slot);
                    }
                chk[t] = c;
            });
        for (auto& x : th) x.join();
        double mean = 0, mx = 0;
        for (double v : ns) {
            mean += v / T;
            mx = std::max(mx, v);
        }
        printf("K=%d variant %d (%s): %.1f ns per push (slowest thread %.1f), layout checksum %016llx\n", K, variant,
               variant == 0 ? "16-byte entries" : variant == 1 ? "f64 scores + u32 slots" : variant == 2 ? "f64 scores + u16 slots" : variant == 3 ? "16-byte entries, huge pages" : variant == 4 ? "16-byte entries, integer compares" : variant == 5 ? "integer compares, lockstep branch-free climbs" : variant == 6 ? "integer compares, unconditional levels" : variant == 7 ? "integer compares, top-down early stop" : variant == 8 ? "integer compares, branch-free last levels" : "integer compares, payload log", mean, mx,
               (unsigned long long)chk[0]);
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16;
    const int N = argc > 2 ? atoi(argv[2]) : 10001;
    const int pushes = argc > 3 ? atoi(argv[3]) : 300000;
    run<7>(T, N, pushes);
    run<6>(T, N, pushes);
    run<1>(T, N, pushes);
    return 0;
}
