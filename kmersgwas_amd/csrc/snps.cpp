// snps.cpp — kgwas_snps_*: the SNP twin of the association scan (the reference's
// MultipleSNPsDataBases, src/snps_multiple_databases.{h,cpp}, driven by src/associate_snps.cpp).
//
// open   : .fam names (first space-separated token per line, :181-194), sample -> (byte, shift) map (:204-219),
//          .bed size guards with the reference's messages (:81-92), whole .bed in host memory;
// scores : per chunk of SNPs the raw bytes go to the GPU, snp_planes_kernel builds the three bit planes, snp_score_kernel
//          gives calculate_grammmar_approx_association for every phenotype column (bit-identical);
// best   : get_most_associated_snps (:229-241): add_association(0, score, snp) in SNP order into a
//          BestAssociationsHeap per column, then get_rows_sorted_indices (src/best_associations_heap.cpp:135-147);
// write  : output_plink_bed_file (:252-286): the chosen SNPs' .bim lines and .bed bytes, file order.
// No CPU fallback: scoring needs the GPU.
#include <algorithm>
#include <atomic>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "common.h"
#include "heap.h"
#include "ingest.h"
#include "kernels.h"

struct kgwas_snps {
    std::string base;
    std::vector<std::string> samples;
    uint64_t n_samples_file = 0, n_snps = 0, bytes_per_snp = 0;
    std::vector<uint32_t> byte_idx, shift;
    std::vector<uint8_t> bed;  // body of the .bed (without the 3 magic bytes)
};

namespace kgwas {
unsigned usable_cpus();  // scan_host.cpp: cgroup quota / affinity mask
}
using namespace kgwas;

namespace {

template <class T>
struct DevArr {
    T* p = nullptr;
    void alloc(size_t n) { KGWAS_HIP(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T))); }
    ~DevArr() {
        if (p) (void)hipFree(p);
    }
};

void need_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(KGWAS_ERR_DEVICE, "no HIP device available: libkgwas has no CPU fallback");
    KGWAS_HIP(hipSetDevice(device));
}

// Scores of SNPs [first, first + c) for all columns into out[p * out_stride + i], i = 0..c-1.
struct SnpScorer {
    const kgwas_snps* s;
    uint64_t S, P, W_m, L, ndw, chunk;
    DevArr<uint8_t> d_bed;
    DevArr<uint32_t> d_bidx, d_shift, d_planes;
    DevArr<float> d_Y;
    DevArr<double> d_scores;
    hipStream_t st = nullptr;
    SnpScorer(const kgwas_snps* s_, const float* Y, uint64_t n_pheno) : s(s_) {
        S = s->samples.size();
        P = n_pheno;
        W_m = 2 * ((S + 127) / 128);  // m_uint64_words (:71)
        L = 64 * W_m;
        ndw = 2 * W_m;
        chunk = std::max<uint64_t>(1024, std::min<uint64_t>(1u << 20, (512ull << 20) / std::max<uint64_t>(8 * P, s->bytes_per_snp)));
        // permute_scores (src/kmer_general.cpp:155-167) of the zero-padded column: R[128b+4s+l] = V[128b+32l+31-s]
        std::vector<float> Yperm(P * L, 0.0f), V(L);
        for (uint64_t j = 0; j < P; j++) {
            std::fill(V.begin(), V.end(), 0.0f);
            for (uint64_t i = 0; i < S; i++) V[i] = Y[j * S + i];
            for (uint64_t b = 0; b < L / 128; b++)
                for (uint64_t sx = 0; sx < 32; sx++)
                    for (uint64_t l = 0; l < 4; l++) Yperm[j * L + 128 * b + 4 * sx + l] = V[128 * b + 32 * l + 31 - sx];
        }
        d_bed.alloc(chunk * s->bytes_per_snp);
        d_bidx.alloc(S);
        d_shift.alloc(S);
        d_planes.alloc(chunk * 3 * ndw);
        d_Y.alloc(P * L);
        d_scores.alloc(P * chunk);
        KGWAS_HIP(hipMemcpy(d_bidx.p, s->byte_idx.data(), S * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(d_shift.p, s->shift.data(), S * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(d_Y.p, Yperm.data(), Yperm.size() * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));  // (last: a constructor that throws runs no destructor)
    }
    ~SnpScorer() {
        if (st) (void)hipStreamDestroy(st);
    }
    void run(uint64_t first, uint64_t c, double mac, double* out, uint64_t out_stride) {
        KGWAS_HIP(hipMemcpyAsync(d_bed.p, s->bed.data() + first * s->bytes_per_snp, c * s->bytes_per_snp, hipMemcpyHostToDevice, st));
        KGWAS_HIP(launch_snp_planes(d_bed.p, c, (uint32_t)s->bytes_per_snp, d_bidx.p, d_shift.p, (uint32_t)S, (uint32_t)ndw,
                                    d_planes.p, st));
        KGWAS_HIP(launch_snp_score(d_planes.p, c, (uint32_t)ndw, d_Y.p, (uint32_t)L, (uint32_t)P, mac, d_scores.p, st));
        if (out_stride == c)  // [column][SNP of the chunk], as on the device: one transfer
            KGWAS_HIP(hipMemcpyAsync(out, d_scores.p, P * c * sizeof(double), hipMemcpyDeviceToHost, st));
        else
            for (uint64_t j = 0; j < P; j++)
                KGWAS_HIP(hipMemcpyAsync(out + j * out_stride, d_scores.p + j * c, c * sizeof(double), hipMemcpyDeviceToHost, st));
        KGWAS_HIP(hipStreamSynchronize(st));
    }
};

}  // namespace

extern "C" {

int kgwas_snps_open(const char* base_bedbim, const char* const* samples, uint64_t n_samples, kgwas_snps** out) {
    return guarded([&] {
        if (!base_bedbim || (!samples && n_samples) || !out) throw Error(KGWAS_ERR_ARG, "kgwas_snps_open: null argument");
        std::unique_ptr<kgwas_snps> s(new kgwas_snps);
        s->base = base_bedbim;
        for (uint64_t i = 0; i < n_samples; i++) s->samples.push_back(samples[i]);
        // get_names_from_fam_file (:181-194)
        std::vector<std::string> all;
        {
            std::ifstream fin(s->base + ".fam");
            std::string line, cell;
            while (std::getline(fin, line)) {
                std::stringstream ls(line);
                std::vector<std::string> toks;
                while (std::getline(ls, cell, ' ')) toks.push_back(cell);
                if (toks.empty()) toks.push_back("");
                all.push_back(toks[0]);
            }
        }
        // create_map_from_all_samples (:204-219)
        for (const std::string& name : s->samples) {
            const size_t i_full = std::find(all.begin(), all.end(), name) - all.begin();
            if (i_full == all.size()) throw Error(KGWAS_ERR_FORMAT, "All accessions should be in fam file: " + name);
            s->byte_idx.push_back((uint32_t)(i_full / 4));
            s->shift.push_back((uint32_t)((i_full % 4) * 2));
        }
        std::ifstream bed(s->base + ".bed", std::ios::binary | std::ios::ate);
        if (!bed) throw Error(KGWAS_ERR_IO, "Couldn't open bed file: " + s->base + ".bed");
        const uint64_t size = (uint64_t)bed.tellg();
        bed.seekg(0, std::ios::beg);
        if (size < 3) throw Error(KGWAS_ERR_FORMAT, "Bed file is too small");  // :85-86
        s->n_samples_file = all.size();
        s->bytes_per_snp = (4 + s->n_samples_file - 1) / 4;
        if (s->bytes_per_snp == 0) throw Error(KGWAS_ERR_FORMAT, "Ilegal size of bed file");
        s->n_snps = (size - 3) / s->bytes_per_snp;
        if (size != s->n_snps * s->bytes_per_snp + 3) throw Error(KGWAS_ERR_FORMAT, "Ilegal size of bed file");  // :91-92
        std::cerr << s->base << "\t(snps,samples) = " << s->n_snps << ", " << s->n_samples_file << std::endl;
        bed.ignore(3);
        s->bed.resize(size - 3);
        bed.read(reinterpret_cast<char*>(s->bed.data()), (std::streamsize)(size - 3));
        if ((uint64_t)bed.gcount() != size - 3) throw Error(KGWAS_ERR_IO, "read error on " + s->base + ".bed");
        *out = s.release();
    });
}

int kgwas_snps_info(const kgwas_snps* s, uint64_t* n_snps, uint64_t* n_samples_file, uint64_t* bytes_per_snp) {
    return guarded([&] {
        if (!s) throw Error(KGWAS_ERR_ARG, "kgwas_snps_info: null");
        if (n_snps) *n_snps = s->n_snps;
        if (n_samples_file) *n_samples_file = s->n_samples_file;
        if (bytes_per_snp) *bytes_per_snp = s->bytes_per_snp;
    });
}

int kgwas_snps_scores(kgwas_snps* s, const float* Y, uint64_t n_pheno, double mac, int device, double* scores) {
    return guarded([&] {
        if (!s || (n_pheno && (!Y || !scores))) throw Error(KGWAS_ERR_ARG, "kgwas_snps_scores: null argument");
        need_device(device);
        if (n_pheno == 0 || s->n_snps == 0) return;
        SnpScorer sc(s, Y, n_pheno);
        for (uint64_t pos = 0; pos < s->n_snps; pos += sc.chunk) {
            const uint64_t c = std::min<uint64_t>(sc.chunk, s->n_snps - pos);
            sc.run(pos, c, mac, scores + pos, s->n_snps);
        }
    });
}

int kgwas_snps_best(kgwas_snps* s, const float* Y, uint64_t n_pheno, uint64_t topn, double mac, int device, uint64_t* counts,
                    uint64_t* indices) {
    return guarded([&] {
        if (!s || (n_pheno && (!Y || !counts || !indices))) throw Error(KGWAS_ERR_ARG, "kgwas_snps_best: null argument");
        need_device(device);
        require_heap_emulation();
        std::vector<BestHeap> heaps;
        for (uint64_t j = 0; j < n_pheno; j++) heaps.emplace_back((size_t)topn);
        if (n_pheno && s->n_snps) {
            SnpScorer sc(s, Y, n_pheno);
            PinBuf<double> buf;  // (pinned: the chunk's scores arrive at the link's rate, not through a staging copy)
            buf.alloc(n_pheno * sc.chunk);
            // every column's heap is its own: the columns of a chunk are offered on a few threads (one thread pushing 2 M SNPs x
            // 101 columns through their heaps was 0.8 of the second the whole call took)
            const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(usable_cpus(), 16), n_pheno));
            for (uint64_t pos = 0; pos < s->n_snps; pos += sc.chunk) {
                const uint64_t c = std::min<uint64_t>(sc.chunk, s->n_snps - pos);
                sc.run(pos, c, mac, buf.p, c);
                std::atomic<uint64_t> next(0);
                auto work = [&] {
                    for (uint64_t j; (j = next.fetch_add(1, std::memory_order_relaxed)) < n_pheno;) {
                        BestHeap& h = heaps[j];  // add_association(0, score, snp_i), SNP order (:235-238)
                        const double* sc_j = buf.p + j * c;
                        for (uint64_t i = 0; i < c; i++) h.add(0, sc_j[i], (size_t)(pos + i));
                    }
                };
                kgwas_run_on_threads(nt, "kgwas-snpheap", work);
            }
        }
        for (uint64_t j = 0; j < n_pheno; j++) {  // get_rows_sorted_indices
            const std::vector<uint64_t> r = heaps[j].rows_sorted();
            counts[j] = r.size();
            std::copy(r.begin(), r.end(), indices + j * topn);
        }
    });
}

int kgwas_snps_write(kgwas_snps* s, uint64_t n_lists, const char* const* out_bases, const uint64_t* counts,
                     const uint64_t* indices, uint64_t stride) {
    return guarded([&] {
        if (!s || (n_lists && (!out_bases || !counts || !indices))) throw Error(KGWAS_ERR_ARG, "kgwas_snps_write: null argument");
        // The .bim is read once and its lines indexed; every list then writes its own two files - lines and genotype rows of its
        // SNPs, in order - on one of a few threads. (One pass over the SNPs that asked every list about every SNP and
        // flushed each line took 1.1 s for 101 lists of 10 001 among 2 M SNPs; the bytes are the same.)
        std::string bim_text;
        {
            std::ifstream bim(s->base + ".bim", std::ios::binary | std::ios::ate);
            if (bim) {
                const std::streamsize sz = bim.tellg();
                bim.seekg(0);
                bim_text.resize((size_t)std::max<std::streamsize>(sz, 0));
                bim.read(&bim_text[0], sz);
                bim_text.resize((size_t)std::max<std::streamsize>(bim.gcount(), 0));
            }
        }
        std::vector<size_t> line_at;  // start of line i; one more entry: the end of the text (a missing line reads as empty, as getline on a short file)
        line_at.reserve(s->n_snps + 1);
        for (size_t at = 0; line_at.size() < s->n_snps; ) {
            line_at.push_back(std::min(at, bim_text.size()));
            const size_t nl = at < bim_text.size() ? bim_text.find('\n', at) : std::string::npos;
            at = nl == std::string::npos ? bim_text.size() + 1 : nl + 1;
        }
        auto line_of = [&](uint64_t i, const char*& p, size_t& n) {
            const size_t a0 = line_at[i];
            size_t e = a0 < bim_text.size() ? bim_text.find('\n', a0) : std::string::npos;
            if (e == std::string::npos) e = bim_text.size();
            p = bim_text.data() + std::min(a0, bim_text.size());
            n = e > a0 ? e - a0 : 0;
        };
        std::atomic<uint64_t> next(0);
        auto work = [&] {
            try {
                std::string out;
                for (uint64_t l; (l = next.fetch_add(1, std::memory_order_relaxed)) < n_lists;) {
                    const std::string b(out_bases[l]);  // BedBimFilesHandle (src/kmer_general.h:133-147)
                    std::ofstream bed(b + ".bed", std::ios::binary), bimo(b + ".bim", std::ios::binary);
                    if (!bed || !bimo) throw Error(KGWAS_ERR_IO, "cannot create " + b + ".bed/.bim");
                    bed << (char)0x6C << (char)0x1B << (char)0x01;
                    out.clear();
                    uint64_t prev = 0;
                    bool first = true;
                    for (uint64_t q = 0; q < counts[l]; q++) {
                        const uint64_t i = indices[l * stride + q];
                        // (the one-pass form took a list's entries in ascending SNP order and stopped at the first one out of order
                        // or past the end)
                        if (i >= s->n_snps || (!first && i <= prev)) break;
                        first = false;
                        prev = i;
                        const char* p;
                        size_t n;
                        line_of(i, p, n);
                        out.append(p, n);
                        out.push_back('\n');
                        bed.write(reinterpret_cast<const char*>(s->bed.data() + i * s->bytes_per_snp), (std::streamsize)s->bytes_per_snp);
                    }
                    bimo.write(out.data(), (std::streamsize)out.size());
                    bed.flush();
                    bimo.flush();
                    if (!bed || !bimo) throw Error(KGWAS_ERR_IO, "error writing " + b + ".bed/.bim");
                }
            } catch (...) {
                next.store(n_lists);  // (the other threads stop at their next list)
                throw;
            }
        };
        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(usable_cpus(), 16), n_lists));
        kgwas_run_on_threads(nt, "kgwas-snpwrite", work);
    });
}

void kgwas_snps_close(kgwas_snps* s) { delete s; }

}  // extern "C"
