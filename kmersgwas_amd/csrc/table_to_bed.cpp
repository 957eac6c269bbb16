// table_to_bed.cpp — kgwas_table_to_bed: the reference's kmers_table_to_bed (src/kmers_table_to_bed.cpp:93-129) on the GPU.
//
// A batch is up to `batch_size` KEPT k-mers (load_kmers counts the rows it keeps, src/kmers_multiple_databases.cpp:110-113)
// and exists iff table rows were left when it started; every batch gets <base>.<i>.bed/.bim/.fam. With
// unique_patterns only the first k-mer (file order, across all batches) of every presence/absence hash is written
// (:254-264). The device does the per-row work on chunks of the table (squeeze to phenotype order, masked popcount,
// pattern hash, PLINK bytes); the host walks the rows in file order, applies the MAC filter and the hash set, and
// writes. No CPU fallback: the per-row work needs the GPU.
#include <fstream>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <exception>
#include <future>
#include <string>
#include <unordered_set>
#include <vector>

#include "common.h"
#include "ingest.h"
#include "kernels.h"

namespace kgwas {
namespace {

template <class T>
struct Dev {
    T* p = nullptr;
    void alloc(size_t n) { KGWAS_HIP(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T))); }
    ~Dev() {
        if (p) (void)hipFree(p);
    }
};
}  // namespace
}  // namespace kgwas

using namespace kgwas;

extern "C" int kgwas_table_to_bed(kgwas_table* t, const uint64_t* col, uint64_t n_acc, const char* const* acc_names,
                                  const float* y, uint64_t min_count, uint64_t batch_size, int unique_patterns,
                                  const char* out_base, int device, uint64_t* n_batches, uint64_t* n_written) {
    return guarded([&] {
        if (!t || !col || !acc_names || !y || !out_base) throw Error(KGWAS_ERR_ARG, "kgwas_table_to_bed: null argument");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw Error(KGWAS_ERR_DEVICE, "no HIP device available: libkgwas has no CPU fallback");
        KGWAS_HIP(hipSetDevice(device));
        uint64_t S_f = 0, n_rows = 0, W_f = 0;
        uint32_t klen = 0;
        if (kgwas_table_info(t, &S_f, &n_rows, &W_f, &klen) != KGWAS_OK) throw Error(KGWAS_ERR_ARG, kgwas_last_error());
        const uint64_t S = n_acc;
        for (uint64_t i = 0; i < S; i++)
            if (col[i] >= S_f) throw Error(KGWAS_ERR_ARG, "kgwas_table_to_bed: column index out of range");
        const uint32_t W_m = (uint32_t)(2 * ((S + 127) / 128));  // m_hash_words rounded to the 128-bit unit (:51)
        const uint32_t bpr = (uint32_t)((S + 3) / 4);            // write_PA: one byte per 4 accessions
        const uint64_t stride = 1 + W_f;
        const uint64_t piece = std::max<uint64_t>(1024, std::min<uint64_t>(1u << 20, (256ull << 20) / (8 * stride)));

        std::vector<uint32_t> colmap(64ull * W_m, 0xFFFFFFFFu);
        for (uint64_t i = 0; i < S; i++) colmap[i] = (uint32_t)col[i];
        Dev<uint32_t> d_colmap, d_sq, d_n1;
        Dev<uint64_t> d_rows, d_hash;
        Dev<uint8_t> d_bed;
        // Two sets of host buffers: piece k + 1 is read, copied and processed on the GPU (this thread) while piece k's kept rows
        // are written (the writer thread below: batches, the pattern set and the files are its alone, pieces in order).
        struct HostSet {
            PinBuf<uint64_t> rows, hash;
            PinBuf<uint32_t> n1;
            PinBuf<uint8_t> bed;
        } hs[2];
        d_colmap.alloc(colmap.size());
        KGWAS_HIP(hipMemcpy(d_colmap.p, colmap.data(), colmap.size() * 4, hipMemcpyHostToDevice));
        d_rows.alloc(piece * stride);
        d_sq.alloc(piece * 2 * W_m);
        d_n1.alloc(piece);
        d_hash.alloc(piece);
        d_bed.alloc(piece * bpr);
        for (HostSet& h : hs) {
            h.rows.alloc(piece * stride);
            h.n1.alloc(piece);
            h.hash.alloc(piece);
            h.bed.alloc(piece * bpr);
        }
        hipStream_t st = nullptr;
        KGWAS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        struct StreamGuard {
            hipStream_t s;
            ~StreamGuard() { (void)hipStreamDestroy(s); }
        } sg{st};

        std::unordered_set<uint64_t> seen;  // pa_patterns_counter (src/kmers_table_to_bed.cpp:107-108)
        std::ofstream bed, bim;
        bool open = false;
        uint64_t batch = 0, kept = 0, written = 0;
        std::future<void> bed_pending;  // the .bed bytes of the last run, written beside the formatting of the next run's .bim lines
        auto bed_wait = [&] {
            if (bed_pending.valid()) bed_pending.get();
        };
        auto close_batch = [&] {
            bed_wait();
            bed.close();
            bim.close();
            const std::string base = std::string(out_base) + "." + std::to_string(batch);
            if (!bed || !bim) throw Error(KGWAS_ERR_IO, "error writing " + base + ".bed/.bim");  // (a full disk shows here at the latest)
            std::ofstream fam(base + ".fam");  // src/kmers_table_to_bed.cpp:119-124
            if (!fam) throw Error(KGWAS_ERR_IO, "cannot create " + base + ".fam");
            for (uint64_t i = 0; i < S; i++) fam << acc_names[i] << " " << acc_names[i] << " 0 0 0 " << y[i] << std::endl;
            open = false;
            batch++;
            kept = 0;
        };
        // what the writer does with one piece: its kept rows go out in runs - consecutive rows' .bed bytes are adjacent, their
        // .bim lines are put together in one buffer
        auto write_piece = [&](const HostSet& h, uint64_t c) {
            std::string bim_buf;
            bim_buf.reserve((size_t)std::min<uint64_t>(c, 1u << 20) * (klen + 12));
            uint64_t run0 = 0, run_n = 0;  // rows [run0, run0 + run_n) of the piece: kept, not yet in `runs`
            std::vector<std::pair<const char*, std::streamsize>> runs;  // .bed bytes to write, in order
            auto end_run = [&] {
                if (run_n) runs.emplace_back(reinterpret_cast<const char*>(h.bed.p + run0 * bpr), (std::streamsize)(run_n * bpr));
                run_n = 0;
            };
            auto flush_run = [&] {  // everything collected so far goes out: the .bed runs on a thread of their own, the .bim lines here
                end_run();
                if (!runs.empty()) {
                    bed_wait();
                    std::ofstream* bo = &bed;
                    bed_pending = std::async(std::launch::async, [bo, rs = std::move(runs)] {
                        for (const auto& pr : rs) bo->write(pr.first, pr.second);
                    });
                    runs.clear();
                }
                if (!bim_buf.empty()) bim.write(bim_buf.data(), (std::streamsize)bim_buf.size());
                bim_buf.clear();
            };
            for (uint64_t r = 0; r < c; r++) {
                if (!open) {  // a batch starts with the first row read in it
                    const std::string base = std::string(out_base) + "." + std::to_string(batch);
                    bed.open(base + ".bed", std::ios::binary);
                    bim.open(base + ".bim");
                    if (!bed || !bim) throw Error(KGWAS_ERR_IO, "cannot create " + base + ".bed/.bim");
                    bed << (char)0x6C << (char)0x1B << (char)0x01;  // BedBimFilesHandle, src/kmer_general.h:138
                    open = true;
                }
                const uint64_t n1 = h.n1.p[r];
                if (S >= min_count && n1 >= min_count && n1 <= S - min_count) {
                    kept++;
                    if (!unique_patterns || seen.insert(h.hash.p[r]).second) {
                        if (run_n && run0 + run_n != r) end_run();
                        if (!run_n) run0 = r;
                        run_n++;
                        bim_buf += "0\t";
                        {
                            const uint64_t w = h.rows.p[r * stride];
                            char km[32];
                            for (size_t i = 0; i < klen; i++) km[i] = "ACGT"[(w >> (2 * (klen - 1 - i))) & 3];  // bits2kmer31, src/kmer_general.cpp:77-87
                            bim_buf.append(km, klen);
                        }
                        bim_buf += "\t0\t0\t0\t1\n";
                        written++;
                        if (bim_buf.size() > (8u << 20)) flush_run();
                    }
                    if (kept >= batch_size) {  // load_kmers stops reading once the batch is full
                        flush_run();
                        close_batch();
                    }
                }
            }
            if (open) flush_run();
            bed_wait();  // (the piece's buffers go back to the reader)
        };
        const uint64_t n_pieces = (n_rows + piece - 1) / piece;
        std::mutex mu;
        std::condition_variable cv;
        uint64_t ready = 0, done = 0;  // pieces handed to the writer / written
        bool stop = false;
        std::exception_ptr werr;
        std::thread writer([&] {
            try {
                for (uint64_t k = 0; k < n_pieces; k++) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || ready > k; });
                        if (ready <= k) return;
                    }
                    write_piece(hs[k & 1], std::min<uint64_t>(piece, n_rows - k * piece));
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        done = k + 1;
                    }
                    cv.notify_all();
                }
                if (open) close_batch();
            } catch (...) {
                std::unique_lock<std::mutex> lk(mu);
                werr = std::current_exception();
                stop = true;
                cv.notify_all();
            }
        });
        struct WriterJoin {
            std::thread& t;
            std::mutex& mu;
            std::condition_variable& cv;
            bool& stop;
            bool orderly = false;
            ~WriterJoin() {
                if (!orderly) {
                    std::unique_lock<std::mutex> lk(mu);
                    stop = true;
                }
                cv.notify_all();
                if (t.joinable()) t.join();
            }
        } wj{writer, mu, cv, stop};
        for (uint64_t k = 0; k < n_pieces; k++) {
            const uint64_t pos = k * piece, c = std::min<uint64_t>(piece, n_rows - pos);
            HostSet& h = hs[k & 1];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || done + 2 > k; });  // the writer is through with piece k - 2 (this set)
                if (stop) break;
            }
            if (kgwas_table_read_rows(t, pos, c, h.rows.p) != KGWAS_OK) throw Error(KGWAS_ERR_IO, kgwas_last_error());
            KGWAS_HIP(hipMemcpyAsync(d_rows.p, h.rows.p, c * stride * 8, hipMemcpyHostToDevice, st));
            KGWAS_HIP(launch_squeeze(d_rows.p, stride, c, d_colmap.p, W_m, (uint32_t)W_f, d_sq.p, st));
            KGWAS_HIP(launch_bed_rowinfo(d_sq.p, c, W_m, d_n1.p, d_hash.p, st));
            KGWAS_HIP(launch_bed_bytes(d_sq.p, c, W_m, bpr, d_bed.p, st));
            KGWAS_HIP(hipMemcpyAsync(h.n1.p, d_n1.p, c * 4, hipMemcpyDeviceToHost, st));
            KGWAS_HIP(hipMemcpyAsync(h.hash.p, d_hash.p, c * 8, hipMemcpyDeviceToHost, st));
            KGWAS_HIP(hipMemcpyAsync(h.bed.p, d_bed.p, c * bpr, hipMemcpyDeviceToHost, st));
            KGWAS_HIP(hipStreamSynchronize(st));
            {
                std::unique_lock<std::mutex> lk(mu);
                ready = k + 1;
            }
            cv.notify_all();
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || done >= n_pieces; });
            wj.orderly = !stop;
        }
        if (writer.joinable()) {
            cv.notify_all();
            writer.join();  // (closes the last batch)
        }
        if (werr) std::rethrow_exception(werr);
        if (n_batches) *n_batches = batch;
        if (n_written) *n_written = written;
    });
}
