// ingest.h — chunked, double-buffered host / file ingest shared by the scan and the kinship accumulation
// (replaces the reference's load-a-batch-then-compute loops,
// src/associate_kmers.cpp:104-148 and src/emma_kinship_kmers.cpp:86-99).
//
// Three pinned pieces are filled by producer threads, three device pieces receive them on a copy stream, and the
// consumer works on piece k while pieces k+1 and k+2 are copied / queued for copying and produced: the consumer's
// call is synchronous (a piece's scan + replay), and with only two device pieces the copy of piece k+2 could not be
// queued before it returned - the link idled for the consumer's fixed costs of every piece (43 -> ~50 GB/s). The
// consumer sees the rows in order, so results do not depend on the piece size.
#pragma once
#include <condition_variable>
#include <cstdlib>
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
template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        release();
        if (count) KGWAS_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocMapped));
        n = count;
    }
    T* dev() const {
        T* d = nullptr;
        if (p) KGWAS_HIP(hipHostGetDevicePointer((void**)&d, p, 0));
        return d;
    }
    void release() {
        if (p) (void)hipHostFree(p);
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
    // the work on the piece complete (the device buffer is reused three pieces later: three device pieces are in flight).
    using Consume = std::function<void(const uint64_t*, uint64_t, uint64_t)>;

    ~Ingest() {
        for (auto& e : ev_)
            if (e) (void)hipEventDestroy(e);
        if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    }

    // stride = 64-bit words per row; max_piece_rows bounds a piece (the consumer's own chunk limit).
    void run(uint64_t stride, uint64_t n_rows, uint64_t max_piece_rows, hipStream_t stream, const Fill& fill,
             const Consume& consume) {
        if (n_rows == 0) return;
        if (!piece_rows_) {
            uint64_t pr = (128ull << 20) / (8 * stride);  // 128 MiB pieces (64 MiB: 30 % slower, 256 MiB: no faster)
            if (const char* e = getenv("KGWAS_INGEST_PIECE_ROWS"))
                if (atoll(e) > 0) pr = (uint64_t)atoll(e);
            pr = std::max<uint64_t>(128, std::min<uint64_t>(pr, max_piece_rows) / 128 * 128);
            piece_rows_ = pr;
            for (auto& h : h_) h.alloc(pr * stride);
            for (auto& d : d_) d.alloc(pr * stride);
            KGWAS_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
            for (auto& e : ev_) KGWAS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const uint64_t piece = piece_rows_;
        const uint64_t n_pieces = (n_rows + piece - 1) / piece;
        auto count_of = [&](uint64_t k) { return std::min<uint64_t>(piece, n_rows - k * piece); };

        std::mutex mu;
        std::condition_variable cv;
        // A piece is filled in SUB-PIECES of 16 MiB by several producer threads at once (a single thread tops out near
        // 15 GB/s reading the page cache and 28 GB/s copying memory; one thread per 128 MiB piece, the first version, kept
        // at most three of them busy and delivered 27 / 42 GB/s). Items are handed out in order; a piece may be started
        // while it is less than three pieces ahead of the consumer (three pinned buffers).
        const uint64_t sub_rows = std::max<uint64_t>(128, std::min<uint64_t>(piece, ((16ull << 20) / (8 * stride)) / 128 * 128));
        auto subs_of = [&](uint64_t k) { return (count_of(k) + sub_rows - 1) / sub_rows; };
        uint64_t next_piece = 0, next_sub = 0, consumed = 0;  // next item to produce / pieces whose buffers may be overwritten
        std::vector<uint32_t> left(n_pieces);  // sub-pieces of a piece still to be filled
        for (uint64_t k = 0; k < n_pieces; k++) left[k] = (uint32_t)subs_of(k);
        std::vector<char> done(n_pieces, 0);
        bool stop = false;
        std::string producer_error;
        auto producer_main = [&] {
            try {
                for (;;) {
                    uint64_t k, j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || next_piece >= n_pieces || next_piece < consumed + 3; });
                        if (stop || next_piece >= n_pieces) return;
                        k = next_piece;
                        j = next_sub++;
                        if (next_sub >= subs_of(k)) next_piece++, next_sub = 0;
                    }
                    const uint64_t r0 = j * sub_rows, cnt = std::min<uint64_t>(sub_rows, count_of(k) - r0);
                    fill(h_[k % 3].p + r0 * stride, k * piece + r0, cnt);
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
            // producer threads: a share of the CPUs this process may use (the replay workers of the consumer need the
            // rest; a streamed scan is bound by the producers and the link, not by the replay)
            // (measured on 16 quota CPUs, 40 M rows x 1135 samples from the page cache: 5 producers 41.8 GB/s, 8 35.6, 12 31.7)
            // reading the page cache (the kernel's copy, ~8 GB/s per thread) takes more threads than copying memory:
            // half of the CPUs for a file feed (8 of 16: 40.8 GB/s; 5: 35 GB/s), a third for a memory feed
            uint64_t nt = file_feed_ ? std::max(3u, std::min(8u, producer_cpus_ / 2)) : std::max(3u, std::min(6u, producer_cpus_ / 3));
            if (const char* e = getenv("KGWAS_INGEST_THREADS"))
                if (atoi(e) > 0) nt = (uint64_t)atoi(e);
            uint64_t items = 0;
            for (uint64_t k = 0; k < n_pieces && items < nt; k++) items += subs_of(k);
            for (uint64_t i = 0; i < std::min<uint64_t>(nt, items); i++) producers.emplace_back(producer_main);
        }

        auto compute = [&](uint64_t k) {
            KGWAS_HIP(hipStreamWaitEvent(stream, ev_[k % 3], 0));
            consume(d_[k % 3].p, k * piece, count_of(k));
            {
                std::unique_lock<std::mutex> lk(mu);
                consumed = k + 1;
            }
            cv.notify_all();
        };
        for (uint64_t k = 0; k < n_pieces; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || done[k]; });
                if (!done[k]) throw Error(KGWAS_ERR_IO, producer_error.empty() ? "ingest stopped" : producer_error);
            }
            // device piece k % 3 is free: compute(k - 3) returned two turns ago
            KGWAS_HIP(hipMemcpyAsync(d_[k % 3].p, h_[k % 3].p, count_of(k) * stride * 8, hipMemcpyHostToDevice, copy_stream_));
            KGWAS_HIP(hipEventRecord(ev_[k % 3], copy_stream_));
            if (k >= 2) compute(k - 2);
        }
        if (n_pieces >= 2) compute(n_pieces - 2);
        compute(n_pieces - 1);
    }

private:
    PinBuf<uint64_t> h_[3];
    DevBuf<uint64_t> d_[3];
    hipStream_t copy_stream_ = nullptr;
    hipEvent_t ev_[3] = {nullptr, nullptr, nullptr};
    uint64_t piece_rows_ = 0;

public:
    unsigned producer_cpus_ = 16;  // CPUs the process may use (the owner sets it: cgroup quota / GPUs sharing the host)
    bool file_feed_ = false;       // the next run's fill reads a file (set by the owner before run())
};

}  // namespace kgwas
