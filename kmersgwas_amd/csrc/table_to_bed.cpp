// table_to_bed.cpp — kgwas_table_to_bed: the reference's kmers_table_to_bed (src/kmers_table_to_bed.cpp:93-129) on the GPU.
//
// A batch is up to `batch_size` KEPT k-mers (load_kmers counts the rows it keeps, src/kmers_multiple_databases.cpp:110-113)
// and exists iff table rows were left when it started; every batch gets <base>.<i>.bed/.bim/.fam. With
// unique_patterns only the first k-mer (file order, across all batches) of every presence/absence hash is written
// (:254-264). The device does the per-row work on chunks of the table (squeeze to phenotype order, masked popcount,
// pattern hash, PLINK bytes); the host walks the rows in file order, applies the MAC filter and the hash set, and
// writes. No CPU fallback: the per-row work needs the GPU.
#include <fstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "common.h"
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
template <class T>
struct Pin {
    T* p = nullptr;
    void alloc(size_t n) { KGWAS_HIP(hipHostMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault)); }
    ~Pin() {
        if (p) (void)hipHostFree(p);
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
        Pin<uint64_t> h_rows, h_hash;
        Pin<uint32_t> h_n1;
        Pin<uint8_t> h_bed;
        d_colmap.alloc(colmap.size());
        KGWAS_HIP(hipMemcpy(d_colmap.p, colmap.data(), colmap.size() * 4, hipMemcpyHostToDevice));
        d_rows.alloc(piece * stride);
        d_sq.alloc(piece * 2 * W_m);
        d_n1.alloc(piece);
        d_hash.alloc(piece);
        d_bed.alloc(piece * bpr);
        h_rows.alloc(piece * stride);
        h_n1.alloc(piece);
        h_hash.alloc(piece);
        h_bed.alloc(piece * bpr);
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
        auto close_batch = [&] {
            bed.close();
            bim.close();
            const std::string base = std::string(out_base) + "." + std::to_string(batch);
            std::ofstream fam(base + ".fam");  // src/kmers_table_to_bed.cpp:119-124
            if (!fam) throw Error(KGWAS_ERR_IO, "cannot create " + base + ".fam");
            for (uint64_t i = 0; i < S; i++) fam << acc_names[i] << " " << acc_names[i] << " 0 0 0 " << y[i] << std::endl;
            open = false;
            batch++;
            kept = 0;
        };
        for (uint64_t pos = 0; pos < n_rows; pos += piece) {
            const uint64_t c = std::min<uint64_t>(piece, n_rows - pos);
            if (kgwas_table_read_rows(t, pos, c, h_rows.p) != KGWAS_OK) throw Error(KGWAS_ERR_IO, kgwas_last_error());
            KGWAS_HIP(hipMemcpyAsync(d_rows.p, h_rows.p, c * stride * 8, hipMemcpyHostToDevice, st));
            KGWAS_HIP(launch_squeeze(d_rows.p, stride, c, d_colmap.p, W_m, (uint32_t)W_f, d_sq.p, st));
            KGWAS_HIP(launch_bed_rowinfo(d_sq.p, c, W_m, d_n1.p, d_hash.p, st));
            KGWAS_HIP(launch_bed_bytes(d_sq.p, c, W_m, bpr, d_bed.p, st));
            KGWAS_HIP(hipMemcpyAsync(h_n1.p, d_n1.p, c * 4, hipMemcpyDeviceToHost, st));
            KGWAS_HIP(hipMemcpyAsync(h_hash.p, d_hash.p, c * 8, hipMemcpyDeviceToHost, st));
            KGWAS_HIP(hipMemcpyAsync(h_bed.p, d_bed.p, c * bpr, hipMemcpyDeviceToHost, st));
            KGWAS_HIP(hipStreamSynchronize(st));
            // The piece's kept rows go out in runs: consecutive rows' .bed bytes are adjacent in h_bed, their .bim lines are put
            // together in one buffer (one ofstream call per row and field was most of the tool's time: 1.8 GB/s of files).
            std::string bim_buf;
            bim_buf.reserve((size_t)std::min<uint64_t>(c, 1u << 20) * (klen + 12));
            uint64_t run0 = 0, run_n = 0;  // rows [run0, run0 + run_n) of the piece: kept and not yet written
            auto flush_run = [&] {
                if (run_n) bed.write(reinterpret_cast<const char*>(h_bed.p + run0 * bpr), (std::streamsize)(run_n * bpr));
                run_n = 0;
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
                const uint64_t n1 = h_n1.p[r];
                if (S >= min_count && n1 >= min_count && n1 <= S - min_count) {
                    kept++;
                    if (!unique_patterns || seen.insert(h_hash.p[r]).second) {
                        if (run_n && run0 + run_n != r) flush_run();
                        if (!run_n) run0 = r;
                        run_n++;
                        bim_buf += "0\t";
                        {
                            const uint64_t w = h_rows.p[r * stride];
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
        }
        if (open) close_batch();
        if (n_batches) *n_batches = batch;
        if (n_written) *n_written = written;
    });
}
