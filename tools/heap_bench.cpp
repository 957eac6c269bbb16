// heap_bench.cpp — host-side microbenchmark for the replay's inner loop (not part of the product).
// T threads x H heaps of N entries each; every heap receives a stream of "effective pushes" (score above the
// current minimum), as in the ramp part of a pass. Variants:
//   0: std::pop_heap/push_heap on 16-byte entries (what heap.h does)
//   1: hand-written replace with the same element moves (branchless child choice)
//   2: variant 1, the thread's H heaps advanced in lockstep (memory-level parallelism across columns)
// build: g++ -O3 -std=c++17 -pthread tools/heap_bench.cpp -o tools/bin/heap_bench
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <random>
#include <thread>
#include <vector>

struct Ent {
    double score;
    uint32_t slot;
};
struct Greater {
    bool operator()(const Ent& l, const Ent& r) const { return l.score > r.score; }
};

static inline void replace_std(std::vector<Ent>& v, Ent x) {
    std::pop_heap(v.begin(), v.end(), Greater());
    x.slot = v.back().slot;
    v.pop_back();
    v.push_back(x);
    std::push_heap(v.begin(), v.end(), Greater());
}

// Same moves as pop_heap (libstdc++ __adjust_heap + __push_heap) followed by push_heap of x.
static inline void replace_hand(Ent* a, ptrdiff_t n, Ent x) {
    const Ent value = a[n - 1];
    x.slot = a[0].slot;
    const ptrdiff_t len = n - 1;
    ptrdiff_t hole = 0, child = 0;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        child -= (a[child].score > a[child - 1].score) ? 1 : 0;
        a[hole] = a[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1];
        hole = child - 1;
    }
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > 0 && a[parent].score > value.score) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = value;
    hole = n - 1;
    parent = (hole - 1) / 2;
    while (hole > 0 && a[parent].score > x.score) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = x;
}

// variant 3: like replace_hand, the array based at 16 mod 64 bytes (sibling pairs never straddle a line, the four
// grandchildren of a node share one line) and the grandchildren line prefetched one level ahead.
static inline void replace_pf(Ent* a, ptrdiff_t n, Ent x) {
    const Ent value = a[n - 1];
    x.slot = a[0].slot;
    const ptrdiff_t len = n - 1;
    ptrdiff_t hole = 0, child = 0;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        __builtin_prefetch(&a[2 * child - 1]);  // children of (child-1) and of child: 4 consecutive entries
        child -= (a[child].score > a[child - 1].score) ? 1 : 0;
        a[hole] = a[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1];
        hole = child - 1;
    }
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > 0 && a[parent].score > value.score) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = value;
    hole = n - 1;
    parent = (hole - 1) / 2;
    while (hole > 0 && a[parent].score > x.score) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = x;
}

// variant 5: two equal-size heaps advanced in lockstep inside ONE loop (two independent dependency chains per
// iteration); the element moves of each heap are those of replace_hand.
static inline void replace_pair(Ent* a, Ent* b, ptrdiff_t n, Ent xa, Ent xb) {
    const Ent va = a[n - 1], vb = b[n - 1];
    xa.slot = a[0].slot;
    xb.slot = b[0].slot;
    const ptrdiff_t len = n - 1;
    ptrdiff_t ha = 0, ca = 0, hb = 0, cb = 0;
    const ptrdiff_t lim = (len - 1) / 2;
    while (ca < lim && cb < lim) {
        ca = 2 * (ca + 1);
        cb = 2 * (cb + 1);
        ca -= (a[ca].score > a[ca - 1].score) ? 1 : 0;
        cb -= (b[cb].score > b[cb - 1].score) ? 1 : 0;
        a[ha] = a[ca];
        b[hb] = b[cb];
        ha = ca;
        hb = cb;
    }
    while (ca < lim) {
        ca = 2 * (ca + 1);
        ca -= (a[ca].score > a[ca - 1].score) ? 1 : 0;
        a[ha] = a[ca];
        ha = ca;
    }
    while (cb < lim) {
        cb = 2 * (cb + 1);
        cb -= (b[cb].score > b[cb - 1].score) ? 1 : 0;
        b[hb] = b[cb];
        hb = cb;
    }
    if ((len & 1) == 0 && ca == (len - 2) / 2) {
        ca = 2 * (ca + 1);
        a[ha] = a[ca - 1];
        ha = ca - 1;
    }
    if ((len & 1) == 0 && cb == (len - 2) / 2) {
        cb = 2 * (cb + 1);
        b[hb] = b[cb - 1];
        hb = cb - 1;
    }
    ptrdiff_t pa = (ha - 1) / 2, pb = (hb - 1) / 2;
    while (ha > 0 && a[pa].score > va.score) {
        a[ha] = a[pa];
        ha = pa;
        pa = (ha - 1) / 2;
    }
    a[ha] = va;
    while (hb > 0 && b[pb].score > vb.score) {
        b[hb] = b[pb];
        hb = pb;
        pb = (hb - 1) / 2;
    }
    b[hb] = vb;
    ha = n - 1;
    pa = (ha - 1) / 2;
    while (ha > 0 && a[pa].score > xa.score) {
        a[ha] = a[pa];
        ha = pa;
        pa = (ha - 1) / 2;
    }
    a[ha] = xa;
    hb = n - 1;
    pb = (hb - 1) / 2;
    while (hb > 0 && b[pb].score > xb.score) {
        b[hb] = b[pb];
        hb = pb;
        pb = (hb - 1) / 2;
    }
    b[hb] = xb;
}

// variants 6,7,8: K = 3, 4, 6 equal-size heaps in lockstep (generic form of replace_pair); 9,10,11,12: K = 5, 7, 8, 2.
// argv[5]: 1 = prefetch the four grandchildren of the hole before the child is chosen (two levels ahead), 2 = both
// climbs in lockstep and branch-free per heap. Measured on the EPYC 9575F at K = 6/7: neither beats the plain form
// (31-35 ns per push either way), identical layouts.
static int g_pf = 0;
template <int K>
static inline void replace_multi(Ent* const* a, ptrdiff_t n, Ent* x) {
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
    // every hole walks to the leaf level; the walks differ by at most one step (left vs right subtree depth)
    for (;;) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (c[k] < lim) {
                ptrdiff_t cc = 2 * (c[k] + 1);
                if (g_pf) {
                    const ptrdiff_t g = 4 * c[k] + 3;  // grandchildren 4c+3 .. 4c+6: 64 contiguous bytes
                    if (g < len) {
                        __builtin_prefetch(&a[k][g]);
                        __builtin_prefetch(&a[k][g + 3 < len ? g + 3 : len - 1]);
                    }
                }
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
        if (g_pf == 2) continue;
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
    if (g_pf == 2) {  // both climbs in lockstep, branch-free per heap: one loop-exit misprediction per phase, not per heap
        ptrdiff_t hh[K];
#pragma unroll
        for (int k = 0; k < K; k++) hh[k] = h[k];
        for (;;) {
            bool any = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const ptrdiff_t p = hh[k] > 0 ? (hh[k] - 1) / 2 : 0;
                const bool go = (hh[k] > 0) & (a[k][p].score > v[k].score);
                const Ent src = a[k][go ? p : hh[k]];
                a[k][hh[k]] = src;
                hh[k] = go ? p : hh[k];
                any |= go;
            }
            if (!any) break;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            a[k][hh[k]] = v[k];
            hh[k] = n - 1;
        }
        for (;;) {
            bool any = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const ptrdiff_t p = hh[k] > 0 ? (hh[k] - 1) / 2 : 0;
                const bool go = (hh[k] > 0) & (a[k][p].score > x[k].score);
                const Ent src = a[k][go ? p : hh[k]];
                a[k][hh[k]] = src;
                hh[k] = go ? p : hh[k];
                any |= go;
            }
            if (!any) break;
        }
#pragma unroll
        for (int k = 0; k < K; k++) a[k][hh[k]] = x[k];
    }
}

template <int K>
static void run_multi(std::vector<std::vector<Ent>>& heaps, std::vector<std::vector<uint64_t>>& km,
                      std::vector<std::vector<uint64_t>>& rw, const std::vector<double>& u, int H, int N, int pushes) {
    int h = 0;
    for (; h + K <= H; h += K)
        for (int i = 0; i < pushes; i++) {
            Ent* a[K];
            Ent x[K];
            uint32_t sl[K];
            for (int k = 0; k < K; k++) {
                a[k] = heaps[h + k].data();
                const double lo = a[k][0].score;
                x[k] = Ent{lo + (1.0 - lo) * u[(size_t)(h + k) * pushes + i], 0};
                sl[k] = a[k][0].slot;
            }
            replace_multi<K>(a, N, x);
            for (int k = 0; k < K; k++) {
                km[h + k][sl[k]] = i;
                rw[h + k][sl[k]] = i;
            }
        }
    for (; h < H; h++)
        for (int i = 0; i < pushes; i++) {
            Ent* a = heaps[h].data();
            const double lo = a[0].score;
            Ent x{lo + (1.0 - lo) * u[(size_t)h * pushes + i], 0};
            const uint32_t slot = a[0].slot;
            replace_hand(a, N, x);
            km[h][slot] = i;
            rw[h][slot] = i;
        }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16;
    const int H = argc > 2 ? atoi(argv[2]) : 7;
    const int N = argc > 3 ? atoi(argv[3]) : 10001;
    const int pushes = argc > 4 ? atoi(argv[4]) : 400000;
    g_pf = argc > 5 ? atoi(argv[5]) : 0;
    for (int variant = 0; variant < 13; variant++) {
        std::vector<double> ns(T);
        std::vector<uint64_t> chk(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                std::mt19937_64 rng(1234 + t);
                std::uniform_real_distribution<double> U(0.0, 1.0);
                std::vector<std::vector<Ent>> heaps(H);
                std::vector<std::vector<uint64_t>> km(H), rw(H);
                for (int h = 0; h < H; h++) {
                    for (int i = 0; i < N; i++) {
                        heaps[h].push_back(Ent{U(rng), (uint32_t)i});
                        std::push_heap(heaps[h].begin(), heaps[h].end(), Greater());
                    }
                    km[h].resize(N);
                    rw[h].resize(N);
                }
                // effective pushes: min + a random gap distribution similar to an order statistic stream
                std::vector<double> u((size_t)pushes * H);
                for (auto& x : u) x = U(rng);
                auto t0 = std::chrono::steady_clock::now();
                if (variant == 0) {
                    for (int h = 0; h < H; h++)
                        for (int i = 0; i < pushes; i++) {
                            auto& v = heaps[h];
                            const double lo = v.front().score;
                            Ent x{lo + (1.0 - lo) * u[(size_t)h * pushes + i], 0};
                            replace_std(v, x);
                            km[h][v.back().slot] = i;
                            rw[h][v.back().slot] = i;
                        }
                } else if (variant == 1) {
                    for (int h = 0; h < H; h++)
                        for (int i = 0; i < pushes; i++) {
                            Ent* a = heaps[h].data();
                            const double lo = a[0].score;
                            Ent x{lo + (1.0 - lo) * u[(size_t)h * pushes + i], 0};
                            const uint32_t slot = a[0].slot;
                            replace_hand(a, N, x);
                            km[h][slot] = i;
                            rw[h][slot] = i;
                        }
                } else if (variant == 6) {
                    run_multi<3>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 7) {
                    run_multi<4>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 8) {
                    run_multi<6>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 9) {
                    run_multi<5>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 10) {
                    run_multi<7>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 11) {
                    run_multi<8>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 12) {
                    run_multi<2>(heaps, km, rw, u, H, N, pushes);
                } else if (variant == 5) {
                    int h = 0;
                    for (; h + 1 < H; h += 2)
                        for (int i = 0; i < pushes; i++) {
                            Ent* a = heaps[h].data();
                            Ent* b = heaps[h + 1].data();
                            const double la = a[0].score, lb = b[0].score;
                            Ent xa{la + (1.0 - la) * u[(size_t)h * pushes + i], 0};
                            Ent xb{lb + (1.0 - lb) * u[(size_t)(h + 1) * pushes + i], 0};
                            const uint32_t sa = a[0].slot, sb = b[0].slot;
                            replace_pair(a, b, N, xa, xb);
                            km[h][sa] = i;
                            rw[h][sa] = i;
                            km[h + 1][sb] = i;
                            rw[h + 1][sb] = i;
                        }
                    for (; h < H; h++)
                        for (int i = 0; i < pushes; i++) {
                            Ent* a = heaps[h].data();
                            const double lo = a[0].score;
                            Ent x{lo + (1.0 - lo) * u[(size_t)h * pushes + i], 0};
                            const uint32_t slot = a[0].slot;
                            replace_hand(a, N, x);
                            km[h][slot] = i;
                            rw[h][slot] = i;
                        }
                } else if (variant >= 3) {
                    // copy every heap into a buffer based at 16 mod 64
                    std::vector<std::vector<char>> raw(H);
                    std::vector<Ent*> base(H);
                    for (int h = 0; h < H; h++) {
                        raw[h].resize((size_t)N * sizeof(Ent) + 256);
                        uintptr_t p = (uintptr_t)raw[h].data();
                        p = (p + 63) / 64 * 64 + (variant == 3 ? 16 : 0);
                        base[h] = (Ent*)p;
                        for (int i = 0; i < N; i++) base[h][i] = heaps[h][i];
                    }
                    t0 = std::chrono::steady_clock::now();
                    for (int h = 0; h < H; h++)
                        for (int i = 0; i < pushes; i++) {
                            Ent* a = base[h];
                            const double lo = a[0].score;
                            Ent x{lo + (1.0 - lo) * u[(size_t)h * pushes + i], 0};
                            const uint32_t slot = a[0].slot;
                            if (variant == 3) replace_pf(a, N, x); else replace_hand(a, N, x);
                            km[h][slot] = i;
                            rw[h][slot] = i;
                        }
                    for (int h = 0; h < H; h++)
                        for (int i = 0; i < N; i++) heaps[h][i] = base[h][i];
                } else {
                    for (int i = 0; i < pushes; i++)
                        for (int h = 0; h < H; h++) {
                            Ent* a = heaps[h].data();
                            const double lo = a[0].score;
                            Ent x{lo + (1.0 - lo) * u[(size_t)h * pushes + i], 0};
                            const uint32_t slot = a[0].slot;
                            replace_hand(a, N, x);
                            km[h][slot] = i;
                            rw[h][slot] = i;
                        }
                }
                ns[t] = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() /
                        ((double)pushes * H);
                uint64_t c = 0;
                for (int h = 0; h < H; h++)
                    for (int i = 0; i < N; i++) c = c * 1315423911u + heaps[h][i].slot + (uint64_t)(heaps[h][i].score * 1e9);
                chk[t] = c;
            });
        for (auto& x : th) x.join();
        double mx = 0, av = 0;
        for (double x : ns) {
            mx = std::max(mx, x);
            av += x / T;
        }
        printf("variant %d: T=%d H=%d N=%d  ns/push avg %.1f max %.1f  chk %llx\n", variant, T, H, N, av, mx,
               (unsigned long long)chk[0]);
    }
    return 0;
}
