// table.cpp — host-side file formats of the association-scan path: the .table/.names reader,
// the phenotype TSV loader, the PLINK bed/bim/fam writer for the winners and the kinship
// matrix printer. Pure I/O and text; all arithmetic on table bits happens on the GPU.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>
#include <cmath>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <mutex>
#include <thread>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"

namespace kgwas {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
unsigned usable_cpus();  // scan_host.cpp: cgroup quota / affinity mask

}  // namespace kgwas

using namespace kgwas;

struct kgwas_table {
    std::string base;
    int fd = -1;
    uint64_t n_acc = 0, n_rows = 0, W_f = 0;
    uint32_t kmer_len = 0;
    std::vector<std::string> names;
};

struct kgwas_pheno {
    std::vector<std::string> names;  // phenotype column names
    std::vector<std::string> acc;    // accession ids, file order
    std::vector<float> Y;            // n_pheno x n_acc
};

static const uint32_t TABLE_MAGIC = 0xDDCCBBAAu;  // src/kmers_multiple_databases.cpp:78

// load_kmers_talbe_column_names (src/kmer_general.cpp:45-53): whitespace separated tokens.
static std::vector<std::string> read_names(const std::string& base) {
    std::ifstream fin(base + ".names");
    std::vector<std::string> res;
    std::string word;
    while (fin >> word) res.push_back(word);
    return res;
}

namespace {

template <class F>
void parallel_items(unsigned threads, size_t n, const F& fn) {
    if (n == 0) return;
    threads = (unsigned)std::min<size_t>(std::max(1u, threads), n);
    if (threads == 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next(0);
    kgwas_run_on_threads(threads, "kgwas-items", [&] {
        try {
            for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(i);
        } catch (...) {
            next.store(n);  // (the other threads stop at their next item)
            throw;
        }
    });
}

struct FdFile {  // output file written in large pieces; open only while a piece is written
    // (A process that holds hundreds of descriptors at once grows its descriptor table, and in a multi-threaded process the
    // kernel waits for an RCU grace period every time it does - 100 ms and more on a 256-CPU host, found as 0.3 s of
    // "file creation" in the command-line tool's output stage. With a file open only for the duration of a write no more
    // than two descriptors per thread exist at a time.)
    std::string path;
    std::string head;  // written together with the first piece
    bool created = false;
    void create(const std::string& p) { path = p; }
    void append(const char* d, size_t n) {
        std::string h;
        if (!head.empty()) {
            h.swap(head);
            if (n) h.append(d, n);
            d = h.data();
            n = h.size();
        }
        const int fd = ::open(path.c_str(), created ? (O_WRONLY | O_APPEND) : (O_WRONLY | O_CREAT | O_TRUNC), 0666);
        if (fd < 0) throw Error(KGWAS_ERR_IO, "cannot create " + path);
        created = true;
        while (n) {
            const ssize_t w = ::write(fd, d, n);
            if (w <= 0) {
                ::close(fd);
                throw Error(KGWAS_ERR_IO, "write error on " + path);
            }
            d += w;
            n -= (size_t)w;
        }
        ::close(fd);
    }
};

}  // namespace

extern "C" {

const char* kgwas_last_error(void) { return g_last_error.c_str(); }
int kgwas_version(void) { return 100; }

int kgwas_table_open(const char* base, uint32_t kmer_len, kgwas_table** out) {
    return guarded([&] {
        if (!base || !out) throw Error(KGWAS_ERR_ARG, "kgwas_table_open: null argument");
        std::unique_ptr<kgwas_table> t(new kgwas_table);
        t->base = base;
        t->names = read_names(t->base);
        t->n_acc = t->names.size();
        const std::string path = t->base + ".table";
        t->fd = ::open(path.c_str(), O_RDONLY);
        if (t->fd < 0) throw Error(KGWAS_ERR_IO, "Couldn't open kmer table file: " + path);
        struct stat st;
        if (fstat(t->fd, &st) != 0) {
            ::close(t->fd);
            throw Error(KGWAS_ERR_IO, "Couldn't stat kmer table file: " + path);
        }
        const uint64_t fsize = (uint64_t)st.st_size;
        auto fail = [&](const std::string& m) {
            ::close(t->fd);
            t->fd = -1;
            throw Error(KGWAS_ERR_FORMAT, m);
        };
        // Guards of MultipleKmersDataBases' ctor (src/kmers_multiple_databases.cpp:65-93)
        if (fsize <= 16) fail("Kmer table size is too small");
        unsigned char hdr[16];
        if (pread(t->fd, hdr, 16, 0) != 16) fail("Kmer table size is too small");
        uint32_t prefix, k;
        uint64_t nacc;
        memcpy(&prefix, hdr, 4);
        memcpy(&nacc, hdr + 4, 8);
        memcpy(&k, hdr + 12, 4);
        if (prefix != TABLE_MAGIC) fail("Incorrect prefix");
        if (nacc != t->n_acc) fail("Number of accession in file not as defined in class");
        if (kmer_len != 0 && k != kmer_len) fail("Kmer length not as defined in class");
        t->kmer_len = k;
        t->W_f = (t->n_acc + 63) / 64;
        const uint64_t row_bytes = 8 * (1 + t->W_f);
        if ((fsize - 16) % row_bytes != 0) fail("size of file not valid");
        t->n_rows = (fsize - 16) / row_bytes;
        *out = t.release();
    });
}

int kgwas_table_info(const kgwas_table* t, uint64_t* n_acc_file, uint64_t* n_rows, uint64_t* words_per_row,
                     uint32_t* kmer_len) {
    return guarded([&] {
        if (!t) throw Error(KGWAS_ERR_ARG, "kgwas_table_info: null table");
        if (n_acc_file) *n_acc_file = t->n_acc;
        if (n_rows) *n_rows = t->n_rows;
        if (words_per_row) *words_per_row = t->W_f;
        if (kmer_len) *kmer_len = t->kmer_len;
    });
}

int kgwas_table_name(const kgwas_table* t, uint64_t i, const char** name) {
    return guarded([&] {
        if (!t || !name || i >= t->names.size()) throw Error(KGWAS_ERR_ARG, "kgwas_table_name: bad argument");
        *name = t->names[i].c_str();
    });
}

int kgwas_table_column_map(const kgwas_table* t, const char* const* acc, uint64_t n, uint64_t* col_out) {
    return guarded([&] {
        if (!t || (!acc && n) || (!col_out && n)) throw Error(KGWAS_ERR_ARG, "kgwas_table_column_map: null argument");
        for (uint64_t i = 0; i < n; i++) {
            const std::string name(acc[i]);
            uint64_t found = ~0ull;
            for (uint64_t j = 0; j < t->names.size(); j++)
                if (t->names[j] == name) {
                    if (found != ~0ull) throw Error(KGWAS_ERR_FORMAT, "Two DBs with the same name! " + name);
                    found = j;
                }
            if (found == ~0ull) throw Error(KGWAS_ERR_FORMAT, "Couldn't find path for DB: " + name);
            col_out[i] = found;
        }
    });
}

int kgwas_table_read_rows(kgwas_table* t, uint64_t row0, uint64_t n, uint64_t* dst) {
    return guarded([&] {
        if (!t || (!dst && n)) throw Error(KGWAS_ERR_ARG, "kgwas_table_read_rows: null argument");
        if (row0 > t->n_rows || n > t->n_rows - row0) throw Error(KGWAS_ERR_ARG, "kgwas_table_read_rows: out of range");
        const uint64_t row_bytes = 8 * (1 + t->W_f);
        uint64_t off = 16 + row0 * row_bytes, left = n * row_bytes;
        char* p = reinterpret_cast<char*>(dst);
        while (left) {
            ssize_t got = pread(t->fd, p, left > (1ull << 30) ? (1ull << 30) : left, (off_t)off);
            if (got <= 0) throw Error(KGWAS_ERR_IO, "read error on " + t->base + ".table");
            p += got;
            off += (uint64_t)got;
            left -= (uint64_t)got;
        }
    });
}

void kgwas_table_close(kgwas_table* t) {
    if (!t) return;
    if (t->fd >= 0) ::close(t->fd);
    delete t;
}

// load_phenotypes_file (src/kmer_general.cpp:175-205)
int kgwas_pheno_load(const char* path, kgwas_pheno** out) {
    return guarded([&] {
        if (!path || !out) throw Error(KGWAS_ERR_ARG, "kgwas_pheno_load: null argument");
        std::ifstream fin(path);
        if (!fin) throw Error(KGWAS_ERR_IO, std::string("Couldn't open phenotype file: ") + path);
        std::unique_ptr<kgwas_pheno> p(new kgwas_pheno);
        std::vector<std::vector<float>> cols;
        std::string line, cell;
        std::vector<std::string> toks;
        size_t line_n = 0;
        while (std::getline(fin, line)) {
            std::stringstream ls(line);
            toks.clear();
            while (std::getline(ls, cell, '\t')) toks.push_back(cell);
            if (line_n == 0) {
                for (size_t i = 1; i < toks.size(); i++) p->names.push_back(toks[i]);
                cols.resize(p->names.size());
            } else {
                if (toks.size() != p->names.size() + 1)
                    throw Error(KGWAS_ERR_FORMAT,
                                std::string("File should have the same number of fields in each row | ") + path);
                p->acc.push_back(toks[0]);
                for (size_t i = 0; i < p->names.size(); i++) {
                    float v;
                    try {
                        v = std::stof(toks[i + 1]);
                    } catch (const std::exception&) {
                        throw Error(KGWAS_ERR_FORMAT, "stof: bad phenotype value '" + toks[i + 1] + "' in " + path);
                    }
                    cols[i].push_back(v);
                }
            }
            line_n++;
        }
        p->Y.reserve(p->names.size() * p->acc.size());
        for (size_t i = 0; i < cols.size(); i++) p->Y.insert(p->Y.end(), cols[i].begin(), cols[i].end());
        *out = p.release();
    });
}

int kgwas_pheno_info(const kgwas_pheno* p, uint64_t* n_pheno, uint64_t* n_acc) {
    return guarded([&] {
        if (!p) throw Error(KGWAS_ERR_ARG, "kgwas_pheno_info: null");
        if (n_pheno) *n_pheno = p->names.size();
        if (n_acc) *n_acc = p->acc.size();
    });
}
int kgwas_pheno_name(const kgwas_pheno* p, uint64_t j, const char** name) {
    return guarded([&] {
        if (!p || !name || j >= p->names.size()) throw Error(KGWAS_ERR_ARG, "kgwas_pheno_name: bad argument");
        *name = p->names[j].c_str();
    });
}
int kgwas_pheno_accession(const kgwas_pheno* p, uint64_t i, const char** acc) {
    return guarded([&] {
        if (!p || !acc || i >= p->acc.size()) throw Error(KGWAS_ERR_ARG, "kgwas_pheno_accession: bad argument");
        *acc = p->acc[i].c_str();
    });
}
int kgwas_pheno_values(const kgwas_pheno* p, const float** Y) {
    return guarded([&] {
        if (!p || !Y) throw Error(KGWAS_ERR_ARG, "kgwas_pheno_values: null");
        *Y = p->Y.data();
    });
}
void kgwas_pheno_free(kgwas_pheno* p) { delete p; }

uint64_t kgwas_min_count(uint64_t n_acc, double maf, uint64_t mac) {
    // src/associate_kmers.cpp:99-103
    size_t mc = (size_t)std::ceil(static_cast<double>(n_acc) * maf);
    if (mc < mac) mc = mac;
    return mc;
}

// ---- pass 2 of associate_kmers: the winners of ALL phenotype columns -> <base>.bed/.bim/.fam ------------------------------
// The reference re-scans the whole table once per run and writes every column's files from it
// (src/associate_kmers.cpp:167-195). Here the winners' rows are fetched by file row index:
//   1. the union of the columns' winner rows, sorted (a row that wins in several columns is read and expanded once);
//   2. per block of the union: the rows are read by a pool of threads - neighbouring rows in one pread, and, once the
//      reads turn out to be slow (a cold file), with the block's read-ahead requested up front so the device sees a deep
//      queue instead of one request at a time - and expanded to PLINK bytes word-wise (write_PA's two bits per accession:
//      a byte of table bits -> 16 bits through a 256-entry table when the phenotyped accessions are the table's leading
//      columns in order, a gather over precomputed (word, shift) pairs otherwise); the expansion does not depend on the
//      phenotype column;
//   3. per column (in parallel): its winners of the block, in row order, appended to its .bed and .bim.

int kgwas_write_plink_many(uint64_t n_cols, const char* const* out_bases, kgwas_table* t, const uint64_t* col, uint64_t n_acc,
                           const char* const* acc_names, const float* Y, const uint64_t* n_win, const uint64_t* const* kmer_pop,
                           const uint64_t* const* row_pop, uint32_t threads) {
    return guarded([&] {
        if (!t || !col || !acc_names || (n_cols && (!out_bases || !Y || !n_win || !kmer_pop || !row_pop)))
            throw Error(KGWAS_ERR_ARG, "kgwas_write_plink_many: null argument");
        const unsigned T = threads ? threads : usable_cpus();
        const bool trace = opt_set("KGWAS_TRACE");
        double ph[6] = {0, 0, 0, 0, 0, 0};  // sort + union, read-ahead, read + expand, append, fam, output files created
        auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tp = tnow();
        auto lap = [&](int i) {
            const double n = tnow();
            ph[i] += n - tp;
            tp = n;
        };
        const uint64_t W = t->W_f, row_bytes = 8 * (1 + W), bed_bytes = (n_acc + 3) / 4;
        for (uint64_t i = 0; i < n_acc; i++)
            if (col[i] >= t->n_acc) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink_many: column index out of range");
        bool direct = true;
        for (uint64_t i = 0; i < n_acc; i++) direct = direct && col[i] == i;

        // get_kmers_for_output (src/best_associations_heap.cpp:110-127): rank = queue size at pop, then sorted by row
        struct Ent {
            uint64_t row, kmer;
            uint32_t rank;
        };
        std::vector<std::vector<Ent>> lst(n_cols);
        parallel_items(T, n_cols, [&](size_t j) {
            const uint64_t n = n_win[j];
            if (n && (!kmer_pop[j] || !row_pop[j])) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink_many: null winner list");
            if (n > 0xFFFFFFFFull) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink_many: too many winners");
            lst[j].resize(n);
            for (uint64_t i = 0; i < n; i++) {
                if (row_pop[j][i] >= t->n_rows) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink: row index out of range");
                lst[j][i] = Ent{row_pop[j][i], kmer_pop[j][i], (uint32_t)(n - i)};
            }
            std::sort(lst[j].begin(), lst[j].end(), [](const Ent& a, const Ent& b) { return a.row < b.row; });
        });
        // the union of the winner rows: every thread takes a slice of the table's row range, collects the (sorted) columns'
        // entries that fall into it, sorts and de-duplicates them; the slices in order are the union
        std::vector<uint64_t> urow;
        {
            const size_t NB = std::max<size_t>(1, std::min<size_t>(4 * T, 256));
            std::vector<std::vector<uint64_t>> part(NB);
            parallel_items(T, NB, [&](size_t b) {
                const uint64_t lo = (uint64_t)((unsigned __int128)t->n_rows * b / NB), hi = (uint64_t)((unsigned __int128)t->n_rows * (b + 1) / NB);
                std::vector<uint64_t>& v = part[b];
                for (auto& l : lst) {
                    auto cmp = [](const Ent& e, uint64_t r) { return e.row < r; };
                    auto a = std::lower_bound(l.begin(), l.end(), lo, cmp), z = std::lower_bound(l.begin(), l.end(), hi, cmp);
                    for (; a != z; ++a) v.push_back(a->row);
                }
                std::sort(v.begin(), v.end());
                v.erase(std::unique(v.begin(), v.end()), v.end());
            });
            size_t total = 0;
            for (auto& v : part) total += v.size();
            urow.reserve(total);
            for (auto& v : part) urow.insert(urow.end(), v.begin(), v.end());
        }

        lap(0);
        std::vector<FdFile> bed(n_cols), bim(n_cols);
        for (uint64_t j = 0; j < n_cols; j++)
            if (!out_bases[j]) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink_many: null output name");
        for (uint64_t j = 0; j < n_cols; j++) {
            bed[j].create(std::string(out_bases[j]) + ".bed");
            bim[j].create(std::string(out_bases[j]) + ".bim");
            const char magic[3] = {(char)0x6C, (char)0x1B, (char)0x01};  // BedBimFilesHandle, src/kmer_general.h:138
            bed[j].head.assign(magic, 3);
        }

        lap(5);
        // write_PA's byte (src/kmers_multiple_databases.cpp:225-236): accession a of a group of four -> bits 2a, 2a + 1
        uint16_t spread[256];
        for (unsigned b = 0; b < 256; b++) {
            unsigned v = 0;
            for (unsigned k = 0; k < 8; k++)
                if (b & (1u << k)) v |= 3u << (2 * k);
            spread[b] = (uint16_t)v;
        }
        std::vector<uint32_t> g_word(direct ? 0 : n_acc);
        std::vector<uint8_t> g_shift(direct ? 0 : n_acc);
        for (uint64_t i = 0; i < n_acc && !direct; i++) {
            g_word[i] = (uint32_t)(1 + col[i] / 64);
            g_shift[i] = (uint8_t)(col[i] % 64);
        }

        // Winners at most `coalesce` bytes apart are read together. What pays depends on where the file is: from the page
        // cache a read costs a system call (1-3 us, more in sandboxed containers) plus a copy at ~10 GB/s, so skipping over
        // 32 KB of other rows is cheaper than a second call; from a device every 4 KB page read for nothing is bandwidth
        // lost. The first pieces run at 8 KB and time their small reads; the rest of the call uses 32 KB (page cache) or
        // 4 KB + read-ahead hints (cold file). KGWAS_PLINK_COALESCE=bytes: fixed.
        const bool coalesce_fixed = exp_set("KGWAS_PLINK_COALESCE");
        std::atomic<uint64_t> coalesce(coalesce_fixed ? strtoull(exp_str("KGWAS_PLINK_COALESCE"), nullptr, 10) : 8192);
        std::atomic<uint64_t> probe_ns(0), probe_n(0);
        const size_t BLOCK = 1u << 16;  // distinct rows per block
        const uint64_t exp_bytes = (bed_bytes + 15) / 16 * 16 + 16;  // room for the word-wise expansion's overshoot
        std::vector<unsigned char> pa(std::min<size_t>(BLOCK, urow.size()) * exp_bytes);
        std::vector<size_t> cur(n_cols, 0);  // next entry of every column
        std::atomic<int> cold(0);             // reads are slow: ask for the block's read-ahead first
        for (size_t b0 = 0; b0 < urow.size(); b0 += BLOCK) {
            const size_t nb = std::min(BLOCK, urow.size() - b0);
            // -- read + expand, in pieces of 256 union rows
            const size_t PIECE = 256, n_pieces = (nb + PIECE - 1) / PIECE;
            if (cold.load(std::memory_order_relaxed))
                parallel_items(T, n_pieces, [&](size_t pc) {
                    const size_t lo = b0 + pc * PIECE, hi = std::min(b0 + nb, lo + PIECE);
                    const uint64_t COALESCE = coalesce.load(std::memory_order_relaxed);
                    for (size_t u = lo; u < hi;) {
                        size_t v = u + 1;
                        while (v < hi && (urow[v] - urow[v - 1]) * row_bytes <= COALESCE) v++;
                        (void)posix_fadvise(t->fd, (off_t)(16 + urow[u] * row_bytes), (off_t)((urow[v - 1] - urow[u] + 1) * row_bytes), POSIX_FADV_WILLNEED);
                        u = v;
                    }
                });
            lap(1);
            parallel_items(T, n_pieces, [&](size_t pc) {
                const size_t lo = b0 + pc * PIECE, hi = std::min(b0 + nb, lo + PIECE);
                std::vector<uint64_t> buf;
                const uint64_t COALESCE = coalesce.load(std::memory_order_relaxed);
                double small_us = 0;  // time in reads of at most 16 KB: what tells a cold file from the page cache
                size_t n_small = 0;
                for (size_t u = lo; u < hi;) {
                    // neighbouring winners (at most 8 KB of other rows between two of them) in one read
                    size_t v = u + 1;
                    while (v < hi && (urow[v] - urow[v - 1]) * row_bytes <= COALESCE && (urow[v] - urow[u] + 1) * row_bytes <= (1u << 20)) v++;
                    const uint64_t span_rows = urow[v - 1] - urow[u] + 1;
                    buf.resize(span_rows * (1 + W));
                    const size_t want = span_rows * row_bytes;
                    size_t got = 0;
                    const bool timed = b0 == 0 && pc < 8 && want <= 16384;
                    const auto tr0 = timed ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
                    while (got < want) {
                        const ssize_t r = pread(t->fd, (char*)buf.data() + got, want - got, (off_t)(16 + urow[u] * row_bytes + got));
                        if (r <= 0) throw Error(KGWAS_ERR_IO, "read error on " + t->base + ".table");
                        got += (size_t)r;
                    }
                    if (timed) {
                        small_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count();
                        n_small++;
                    }
                    for (size_t x = u; x < v; x++) {
                        const uint64_t* row = buf.data() + (urow[x] - urow[u]) * (1 + W);
                        unsigned char* out = pa.data() + (x - b0) * exp_bytes;
                        if (direct) {
                            const uint64_t nw = (n_acc + 63) / 64;
                            for (uint64_t w = 0; w < nw; w++) {
                                const uint64_t bits = row[1 + w];
                                uint16_t o16[8];
                                for (int k = 0; k < 8; k++) o16[k] = spread[(bits >> (8 * k)) & 0xFF];
                                memcpy(out + 16 * w, o16, 16);
                            }
                            if (n_acc % 4) out[bed_bytes - 1] &= (unsigned char)((1u << (2 * (n_acc % 4))) - 1u);  // accessions beyond n_acc
                        } else {
                            for (uint64_t a0 = 0; a0 < n_acc; a0 += 4) {
                                unsigned b = 0;
                                for (uint64_t k = 0; k < 4 && a0 + k < n_acc; k++)
                                    b |= (unsigned)((row[g_word[a0 + k]] >> g_shift[a0 + k]) & 1ull) * (3u << (2 * k));
                                out[a0 / 4] = (unsigned char)b;
                            }
                        }
                    }
                    u = v;
                }
                if (n_small) {
                    probe_ns.fetch_add((uint64_t)(small_us * 1e3), std::memory_order_relaxed);
                    probe_n.fetch_add(n_small, std::memory_order_relaxed);
                }
            });
            lap(2);
            if (b0 == 0 && !coalesce_fixed) {
                // the first block's probe decides: page cache (a few microseconds per small read) or device (NVMe ~ 80, disks more)?
                const uint64_t n = probe_n.load();
                const double us = n ? (double)probe_ns.load() * 1e-3 / (double)n : 0.0;
                if (n >= 8 && us > 40.0) {
                    cold.store(1);
                    coalesce.store(4096);
                } else {
                    coalesce.store(32768);
                }
            }
            // -- every column appends its winners of this block (row order) to its .bed and .bim
            const uint64_t row_hi = urow[b0 + nb - 1];
            parallel_items(T, n_cols, [&](size_t j) {
                const std::vector<Ent>& l = lst[j];
                size_t c = cur[j];
                if (c >= l.size() || l[c].row > row_hi) return;
                size_t c1 = c;
                while (c1 < l.size() && l[c1].row <= row_hi) c1++;
                const size_t k = t->kmer_len, line_max = 2 + k + 1 + 10 + 9;
                std::vector<char> bedbuf((c1 - c) * bed_bytes), bimbuf((c1 - c) * line_max);
                char* bp = bedbuf.data();
                char* mp = bimbuf.data();
                const uint64_t* ub = urow.data() + b0;
                const uint64_t* ue = ub + nb;
                const uint64_t* at = ub;
                for (; c < c1; c++) {
                    const Ent& e = l[c];
                    at = std::lower_bound(at, ue, e.row);
                    memcpy(bp, pa.data() + (size_t)(at - ub) * exp_bytes, bed_bytes);
                    bp += bed_bytes;
                    // "0\t<kmer>_<rank>\t0\t0\t0\t1\n" (src/kmers_multiple_databases.cpp:219, 245-247; bits2kmer31,
                    // src/kmer_general.cpp:77-87)
                    *mp++ = '0';
                    *mp++ = '\t';
                    uint64_t w = e.kmer;
                    for (size_t i = 0; i < k; i++, w >>= 2) mp[k - 1 - i] = "ACGT"[w & 3u];
                    mp += k;
                    *mp++ = '_';
                    char dig[12];
                    int nd = 0;
                    for (uint32_t v = e.rank; v || !nd; v /= 10) dig[nd++] = (char)('0' + v % 10);
                    while (nd) *mp++ = dig[--nd];
                    memcpy(mp, "\t0\t0\t0\t1\n", 9);
                    mp += 9;
                }
                cur[j] = c;
                bed[j].append(bedbuf.data(), (size_t)(bp - bedbuf.data()));
                bim[j].append(bimbuf.data(), (size_t)(mp - bimbuf.data()));
            });
            lap(3);
        }
        for (uint64_t j = 0; j < n_cols; j++) {  // a column without winners: the magic alone, an empty .bim
            if (!bed[j].created) bed[j].append(nullptr, 0);
            if (!bim[j].created) bim[j].append(nullptr, 0);
        }
        // write_fam_file (src/kmer_general.cpp:207-225)
        parallel_items(T, n_cols, [&](size_t j) {
            std::ostringstream fam;
            const float* y = Y + j * n_acc;
            for (uint64_t i = 0; i < n_acc; i++) fam << acc_names[i] << " " << acc_names[i] << " 0 0 0" << " " << y[i] << "\n";
            FdFile f;
            f.create(std::string(out_bases[j]) + ".fam");
            const std::string s = fam.str();
            f.append(s.data(), s.size());
        });
        lap(4);
        if (trace)
            fprintf(stderr, "[kgwas] write_plink_many: %llu columns, %zu distinct rows, %u threads%s: sort+union %.3f s, files created %.3f, read-ahead %.3f, read+expand %.3f, append %.3f, fam %.3f\n",
                    (unsigned long long)n_cols, urow.size(), T, cold.load() ? " (cold file)" : "", ph[0], ph[5], ph[1], ph[2], ph[3], ph[4]);
    });
}

// Pass 2 of associate_kmers for one phenotype column.
int kgwas_write_plink(const char* out_base, kgwas_table* t, const uint64_t* col, uint64_t n_acc,
                      const char* const* acc_names, const float* y, uint64_t n, const uint64_t* kmer_pop,
                      const uint64_t* row_pop) {
    if (!out_base || !y) {
        set_error("kgwas_write_plink: null argument");
        return KGWAS_ERR_ARG;
    }
    return kgwas_write_plink_many(1, &out_base, t, col, n_acc, acc_names, y, &n, &kmer_pop, &row_pop, 0);
}

// emma_kinship_kmers' output (src/emma_kinship_kmers.cpp:95-111)
uint64_t kgwas_kinship_format(uint64_t n_acc, const uint64_t* K, uint64_t n_used, char* out, uint64_t cap) {
    // `os << v` of the reference is printf's %g (precision 6), which std::to_chars(general, 6) is specified to reproduce; the
    // rows are formatted by a few threads (one ostringstream over 1135 x 1135 cells took 0.2 s, and the tool asked twice).
    const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::min(16u, usable_cpus())), std::max<uint64_t>(n_acc / 32, 1));
    const uint64_t per = (n_acc + T - 1) / T;
    std::vector<std::string> part(T);
    parallel_items(T, T, [&](size_t t) {
        std::string& o = part[t];
        const uint64_t i0 = std::min<uint64_t>(t * per, n_acc), i1 = std::min<uint64_t>(i0 + per, n_acc);
        o.reserve((i1 - i0) * n_acc * 12 + 16);
        char cell[64];
        for (uint64_t i = i0; i < i1; i++) {
            for (uint64_t j = 0; j < n_acc; j++) {
                if (j > 0) o.push_back('\t');
                double v;
                if (i == j)
                    v = 1;
                else {
                    const uint64_t k = (j < i) ? K[i * n_acc + j] : K[j * n_acc + i];
                    v = static_cast<double>(k) / static_cast<double>(n_used);
                }
                const auto r = std::to_chars(cell, cell + sizeof(cell), v, std::chars_format::general, 6);
                o.append(cell, (size_t)(r.ptr - cell));
            }
            o.push_back('\n');
        }
    });
    uint64_t total = 0;
    for (const std::string& o : part) total += o.size();
    if (out && cap) {
        uint64_t at = 0;
        for (const std::string& o : part) {
            if (at >= cap) break;
            const uint64_t n = std::min<uint64_t>(o.size(), cap - at);
            memcpy(out + at, o.data(), n);
            at += n;
        }
    }
    return total;
}

}  // extern "C"
