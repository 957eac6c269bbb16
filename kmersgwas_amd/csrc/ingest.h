// ingest.h — chunked, double-buffered host / file ingest shared by the scan and the kinship accumulation
// (replaces the reference's load-a-batch-then-compute loops,
// src/associate_kmers.cpp:104-148 and src/emma_kinship_kmers.cpp:86-99).
//
// A ring of pinned pieces is filled by producer threads, a ring of device pieces receives them on a copy stream (queued
// by a copier thread as soon as a piece is filled and a device piece is free), and the consumer works on piece k while
// the pieces behind it are copied: its call is synchronous (a piece's scan + replay), and the copies must not wait for
// it (two device pieces: 43 GB/s; three, queued between the consumer's calls: 48; ten and the copier: see run()). A PINNED piece
// is free again when its copy has completed, not when the consumer is through with the rows: the producers run up to
// pinned_pieces_ ahead of the last copy that was queued and wait for the copy event of the piece they overwrite (with
// three pinned pieces tied to the consumer's progress, they could only ever work on ONE piece, and every piece ended
// with most of them idle behind its last sub-piece). The consumer sees the rows in order, so results do not depend
// on any of these sizes.
#pragma once
#include <sys/mman.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace kgwas {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) {
        o.p = nullptr;
        o.n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count) KGWAS_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};
// Pinned, device-mapped host memory. hipHostMalloc costs 0.16 s per GiB (it faults and zeroes 4 KiB pages one by one; a
// fresh `associate_kmers` spent 0.17 of its 0.52 s there): large buffers are mapped with MADV_HUGEPAGE, touched by a few
// threads and registered instead - 5 ms per 512 MiB where transparent huge pages are available (16x), ~45 ms where they
// are not - and fall back to hipHostMalloc if any step fails (KGWAS_PIN_PLAIN=1 forces that).
struct PinRegion {
    void* p = nullptr;       // what the caller uses
    void* map = nullptr;     // the mapping it lies in (registered memory), nullptr: hipHostMalloc
    size_t map_bytes = 0;
    static constexpr size_t HUGE = 2u << 20;
    void alloc(size_t bytes) {
        release();
        if (!bytes) return;
        static const bool plain = opt_set("KGWAS_PIN_PLAIN");
        if (!plain && bytes >= (8u << 20)) {
            const size_t len = (bytes + HUGE - 1) / HUGE * HUGE;
            void* m = mmap(nullptr, len + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
                char* al = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(m) + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
                (void)madvise(al, len, MADV_HUGEPAGE);
                {
                    // touched in segments of 32 MiB by up to 8 threads (this one among them; fewer if no more are to be had)
                    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
                    const size_t seg = 32u << 20, n_seg = (len + seg - 1) / seg;
                    const unsigned nt = (unsigned)std::min<size_t>(std::min<size_t>(8, hw), n_seg);
                    std::atomic<size_t> next(0);
                    kgwas_run_on_threads(nt, "kgwas-touch", [&] {
                        for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n_seg;)
                            for (size_t o = i * seg, b = std::min(len, (i + 1) * seg); o < b; o += 4096) al[o] = 0;
                    });
                }
                if (hipHostRegister(al, len, hipHostRegisterMapped) == hipSuccess) {
                    p = al;
                    map = m;
                    map_bytes = len + HUGE;
                    return;
                }
                (void)hipGetLastError();  // (not sticky: the plain allocation below is the answer)
                munmap(m, len + HUGE);
            }
        }
        KGWAS_HIP(hipHostMalloc(&p, bytes, hipHostMallocMapped));
    }
    void release() {
        if (map) {
            (void)hipHostUnregister(p);
            munmap(map, map_bytes);
        } else if (p)
            (void)hipHostFree(p);
        p = map = nullptr;
        map_bytes = 0;
    }
};
template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    PinRegion r;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    PinBuf(PinBuf&& o) noexcept : p(o.p), n(o.n), r(o.r) {
        o.p = nullptr;
        o.n = 0;
        o.r = PinRegion();
    }
    void alloc(size_t count) {
        release();
        r.alloc(count * sizeof(T));
        p = static_cast<T*>(r.p);
        n = count;
    }
    T* dev() const {
        T* d = nullptr;
        if (p) KGWAS_HIP(hipHostGetDevicePointer((void**)&d, p, 0));
        return d;
    }
    void release() {
        r.release();
        p = nullptr;
        n = 0;
    }
    ~PinBuf() { release(); }
};

class Ingest {
public:
    // fill(dst, row_off, cnt): produce rows [row_off, row_off + cnt) of this feed, in file layout, into pinned memory
    // (runs on a producer thread, up to two pieces ahead of the consumer).
    using Fill = std::function<void(uint64_t*, uint64_t, uint64_t)>;
    // consume(d_rows, row_off, cnt): work on a device piece; `stream` already waits for its copy. Must return with
    // the work on the piece complete (the device buffer is reused device_pieces_ pieces later).
    using Consume = std::function<void(const uint64_t*, uint64_t, uint64_t)>;

    ~Ingest() { drop(); }

    // stride = 64-bit words per row; max_piece_rows bounds a piece (the consumer's own chunk limit).
    void run(uint64_t stride, uint64_t n_rows, uint64_t max_piece_rows, hipStream_t stream, const Fill& fill,
             const Consume& consume) {
        if (n_rows == 0) return;
        if (!piece_rows_) {
            uint64_t pr = (128ull << 20) / (8 * stride);  // 128 MiB pieces (64 MiB: 30 % slower, 256 MiB: no faster)
            if (const char* e = opt_str("KGWAS_INGEST_PIECE_ROWS"))
                if (atoll(e) > 0) pr = (uint64_t)atoll(e);
            pr = std::max<uint64_t>(128, std::min<uint64_t>(pr, max_piece_rows) / 128 * 128);
            if (const char* e = opt_str("KGWAS_INGEST_PINNED"))
                if (atoi(e) >= 2 && atoi(e) <= 16) pinned_pieces_ = (unsigned)atoi(e);
            if (const char* e = opt_str("KGWAS_INGEST_DEVICE"))
                if (atoi(e) >= 2 && atoi(e) <= 64) device_pieces_ = (unsigned)atoi(e);
            try {
                h_.resize(pinned_pieces_);
                ev_h_.assign(pinned_pieces_, nullptr);
                for (auto& h : h_) h.alloc(pr * stride);
                // (blocking: a producer that has to wait for a copy sleeps; the replay workers share its CPU)
                for (auto& e : ev_h_) KGWAS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
                // (the device pieces themselves are allocated below, as many as the feed at hand has pieces: a session that is
                // fed a few small host buffers and then device-resident tables does not hold ten 128 MiB pieces of HBM)
                d_.resize(device_pieces_);
                ev_.assign(device_pieces_, nullptr);
                KGWAS_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
                for (auto& e : ev_) KGWAS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            } catch (...) {
                drop();  // (a later run starts over instead of finding half of the buffers)
                throw;
            }
            piece_rows_ = pr;
        }
        const uint64_t piece = piece_rows_;
        const uint64_t n_pieces = (n_rows + piece - 1) / piece;
        // device pieces for this feed: min(device_pieces_, its pieces), never fewer than an earlier feed had (nothing is in
        // flight between feeds, so the ring may grow here)
        while (device_ready_ < std::min<uint64_t>(device_pieces_, n_pieces)) {
            d_[device_ready_].alloc(piece * stride);
            device_ready_++;
        }
        auto count_of = [&](uint64_t k) { return std::min<uint64_t>(piece, n_rows - k * piece); };

        std::mutex mu;
        std::condition_variable cv;
        // A piece is filled in SUB-PIECES of 4 MiB by several producer threads at once (a single thread tops out near
        // 15 GB/s reading the page cache and 28 GB/s copying memory; one thread per 128 MiB piece, the first version, kept
        // at most three of them busy and delivered 27 / 42 GB/s). Items are handed out in order; a piece may be started
        // once the copy of the piece whose pinned buffer it takes has been queued (its producers then wait for that copy).
        static const uint64_t sub_mib = exp_str("KGWAS_INGEST_SUB_MIB") && atoi(exp_str("KGWAS_INGEST_SUB_MIB")) > 0 ? (uint64_t)atoi(exp_str("KGWAS_INGEST_SUB_MIB")) : 4u;
        const uint64_t sub_rows = std::max<uint64_t>(128, std::min<uint64_t>(piece, ((sub_mib << 20) / (8 * stride)) / 128 * 128));
        const uint64_t NH = pinned_pieces_;
        auto subs_of = [&](uint64_t k) { return (count_of(k) + sub_rows - 1) / sub_rows; };
        uint64_t next_piece = 0, next_sub = 0, queued = 0;  // next item to produce / copies queued
        std::vector<uint32_t> left(n_pieces);  // sub-pieces of a piece still to be filled
        for (uint64_t k = 0; k < n_pieces; k++) left[k] = (uint32_t)subs_of(k);
        std::vector<char> done(n_pieces, 0);
        bool stop = false;
        std::string producer_error, copier_error;
        uint64_t consumed = 0;  // pieces the consumer has returned from
        int dev = 0;
        KGWAS_HIP(hipGetDevice(&dev));
        auto producer_main = [&] {
            kgwas_name_this_thread("kgwas-producer");
            (void)hipSetDevice(dev);
            try {
                for (;;) {
                    uint64_t k, j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || next_piece >= n_pieces || next_piece < queued + NH; });
                        if (stop || next_piece >= n_pieces) return;
                        k = next_piece;
                        j = next_sub++;
                        if (next_sub >= subs_of(k)) next_piece++, next_sub = 0;
                    }
                    // the copy of piece k - NH out of this pinned buffer (queued: see the wait above; the event is recorded
                    // again only for piece k itself, after every one of its sub-pieces is in)
                    if (k >= NH) KGWAS_HIP(hipEventSynchronize(ev_h_[k % NH]));
                    const uint64_t r0 = j * sub_rows, cnt = std::min<uint64_t>(sub_rows, count_of(k) - r0);
                    fill(h_[k % NH].p + r0 * stride, k * piece + r0, cnt);
                    bool last;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        last = --left[k] == 0;
                        if (last) done[k] = 1;
                    }
                    if (last) cv.notify_all();
                }
            } catch (const std::exception& e) {
                std::unique_lock<std::mutex> lk(mu);
                producer_error = e.what();
                stop = true;
                cv.notify_all();
            }
        };
        std::vector<std::thread> producers;
        struct Joiner {  // whatever happens below, the producers are stopped and joined before the buffers go away
            std::vector<std::thread>& t;
            std::mutex& mu;
            std::condition_variable& cv;
            bool& stop;
            ~Joiner() {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    stop = true;
                }
                cv.notify_all();
                for (auto& x : t)
                    if (x.joinable()) x.join();
            }
        } joiner{producers, mu, cv, stop};
        {
            // producer threads: a third of the CPUs this process may use (the replay workers of the consumer need the rest; a
            // streamed scan is bound by the link once the producers deliver ~57 GB/s). Round 4, 16 quota CPUs, 40 M rows x
            // 1024 samples: memory feeds 100 ms with anything from 4 to 12 threads; page-cache feeds (the kernel's copy,
            // ~10 GB/s a thread) 102-105 ms with 5 threads, 101-122 with 8, 117-119 with 10 or 16 - more copying threads only
            // get in each other's and the replay's way. (Round 3 measured 8 threads best for files: the bound then was the
            // record copies' late completion, see scan_gpu.cpp fetch_records, not the producers.)
            uint64_t nt = std::max(3u, std::min(6u, producer_cpus_ / 3));
            if (const char* e = exp_str("KGWAS_INGEST_THREADS"))
                if (atoi(e) > 0) nt = (uint64_t)atoi(e);
            uint64_t items = 0;
            for (uint64_t k = 0; k < n_pieces && items < nt; k++) items += subs_of(k);
            for (uint64_t i = 0; i < std::min<uint64_t>(nt, items); i++) producers.emplace_back(producer_main);
        }

        // The copier: queues piece k's copy as soon as the producers are through with it and device piece k % ND is free (the
        // consumer has returned from piece k - ND). A thread of its own, because the consumer's call is synchronous and the
        // FIRST piece's takes 15 ms at 101 columns (the heaps fill, and most of a scan's pushes belong to its first rows):
        // with the copies queued between the consumer's calls, the link stood still for 14 of those 15 ms - an eighth of a
        // 40 M-row feed. Device pieces are cheap (128 MiB of 288 GB each), so there are enough of them to copy through it.
        const uint64_t ND = device_ready_;  // (>= min(device_pieces_, n_pieces): a feed with fewer pieces than the ring uses each buffer once)
        auto copier_main = [&] {
            kgwas_name_this_thread("kgwas-copier");
            try {
                KGWAS_HIP(hipSetDevice(dev));
                for (uint64_t k = 0; k < n_pieces; k++) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || (done[k] && k < consumed + ND); });
                        if (stop) return;
                    }
                    KGWAS_HIP(hipMemcpyAsync(d_[k % ND].p, h_[k % NH].p, count_of(k) * stride * 8, hipMemcpyHostToDevice, copy_stream_));
                    KGWAS_HIP(hipEventRecord(ev_[k % ND], copy_stream_));
                    KGWAS_HIP(hipEventRecord(ev_h_[k % NH], copy_stream_));
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        queued = k + 1;
                    }
                    cv.notify_all();
                }
            } catch (const std::exception& e) {
                std::unique_lock<std::mutex> lk(mu);
                copier_error = e.what();
                stop = true;
                cv.notify_all();
            }
        };
        producers.emplace_back(copier_main);  // (joined with the producers, whatever happens)

        static const bool trace = exp_set("KGWAS_INGEST_TRACE");  // where the caller's thread spends a run
        double t_wait = 0, t_copy = 0, t_consume = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
        for (uint64_t k = 0; k < n_pieces; k++) {
            {
                const auto t0 = now();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || queued > k; });
                t_wait += ms_since(t0);
                if (queued <= k)
                    throw Error(KGWAS_ERR_IO, !producer_error.empty() ? producer_error : !copier_error.empty() ? copier_error : "ingest stopped");
            }
            KGWAS_HIP(hipStreamWaitEvent(stream, ev_[k % ND], 0));
            if (trace) {
                const auto t0 = now();
                KGWAS_HIP(hipEventSynchronize(ev_[k % ND]));
                t_copy += ms_since(t0);
            }
            const auto t0 = now();
            consume(d_[k % ND].p, k * piece, count_of(k));
            t_consume += ms_since(t0);
            {
                std::unique_lock<std::mutex> lk(mu);
                consumed = k + 1;
            }
            cv.notify_all();
        }
        if (trace)
            fprintf(stderr, "[kgwas ingest] %llu pieces of %llu rows: caller waited %.1f ms for a piece to be queued, %.1f ms for copies, %.1f ms in the consumer\n",
                    (unsigned long long)n_pieces, (unsigned long long)piece, t_wait, t_copy, t_consume);
    }

private:
    void drop() {
        for (auto& e : ev_)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_h_)
            if (e) (void)hipEventDestroy(e);
        ev_.clear();
        ev_h_.clear();
        if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
        copy_stream_ = nullptr;
        h_.clear();
        d_.clear();
        device_ready_ = 0;
        piece_rows_ = 0;
    }
    std::vector<PinBuf<uint64_t>> h_;
    std::vector<hipEvent_t> ev_h_;  // per pinned piece: its copy to the device has completed
    std::vector<DevBuf<uint64_t>> d_;
    hipStream_t copy_stream_ = nullptr;
    std::vector<hipEvent_t> ev_;  // per device piece: its copy has completed
    uint64_t piece_rows_ = 0;

public:
    unsigned producer_cpus_ = 16;  // CPUs the process may use (the owner sets it: cgroup quota / GPUs sharing the host)
    bool file_feed_ = false;       // the next run's fill reads a file (set by the owner before run())
    unsigned pinned_pieces_ = 3;   // pinned pieces the producers fill ahead of the copies
    unsigned device_pieces_ = 10;  // device pieces: copies queued ahead of the consumer (at most; allocated as feeds need them)
    uint64_t device_ready_ = 0;    // device pieces allocated so far
};

}  // namespace kgwas
