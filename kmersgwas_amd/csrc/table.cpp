// table.cpp — host-side file formats of the association-scan path: the .table/.names reader,
// the phenotype TSV loader, the PLINK bed/bim/fam writer for the winners and the kinship
// matrix printer. Pure I/O and text; all arithmetic on table bits happens on the GPU.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "common.h"

namespace kgwas {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

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

static std::string bits_to_kmer(uint64_t w, size_t k) {  // bits2kmer31, src/kmer_general.cpp:77-87
    static const char bp[4] = {'A', 'C', 'G', 'T'};
    std::string s(k, 'X');
    for (size_t i = 0; i < k; i++) {
        s[k - 1 - i] = bp[w & 3u];
        w >>= 2;
    }
    return s;
}

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

// Pass 2 of associate_kmers for one phenotype column.
int kgwas_write_plink(const char* out_base, kgwas_table* t, const uint64_t* col, uint64_t n_acc,
                      const char* const* acc_names, const float* y, uint64_t n, const uint64_t* kmer_pop,
                      const uint64_t* row_pop) {
    return guarded([&] {
        if (!out_base || !t || !col || !acc_names || !y) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink: null argument");
        // get_kmers_for_output (src/best_associations_heap.cpp:110-127): rank = queue size at pop, sort by row
        struct Ent {
            uint64_t kmer, rank, row;
        };
        std::vector<Ent> lst(n);
        for (uint64_t i = 0; i < n; i++) lst[i] = Ent{kmer_pop[i], n - i, row_pop[i]};
        std::sort(lst.begin(), lst.end(), [](const Ent& a, const Ent& b) { return a.row < b.row; });
        const std::string base(out_base);
        std::ofstream bed(base + ".bed", std::ios::binary), bim(base + ".bim", std::ios::out);
        if (!bed || !bim) throw Error(KGWAS_ERR_IO, "cannot create " + base + ".bed/.bim");
        bed << (char)0x6C << (char)0x1B << (char)0x01;  // BedBimFilesHandle, src/kmer_general.h:138
        const uint64_t row_bytes = 8 * (1 + t->W_f);
        std::vector<uint64_t> row(1 + t->W_f);
        std::string bytes;
        for (const Ent& e : lst) {
            if (e.row >= t->n_rows) throw Error(KGWAS_ERR_ARG, "kgwas_write_plink: row index out of range");
            if (pread(t->fd, row.data(), row_bytes, (off_t)(16 + e.row * row_bytes)) != (ssize_t)row_bytes)
                throw Error(KGWAS_ERR_IO, "read error on " + t->base + ".table");
            // write_PA (src/kmers_multiple_databases.cpp:218-239)
            bim << "0\t" << bits_to_kmer(e.kmer, t->kmer_len) << "_" << std::to_string(e.rank) << "\t0\t0\t0\t1\n";
            bytes.clear();
            for (uint64_t a0 = 0; a0 < n_acc; a0 += 4) {
                unsigned char b = 0;
                for (uint64_t k = 0; k < 4 && a0 + k < n_acc; k++) {
                    const uint64_t c = col[a0 + k];
                    if ((row[1 + c / 64] >> (c % 64)) & 1ull) b |= (unsigned char)(3u << (2 * k));
                }
                bytes.push_back((char)b);
            }
            bed.write(bytes.data(), (std::streamsize)bytes.size());
        }
        // write_fam_file (src/kmer_general.cpp:207-225)
        std::ofstream fam(base + ".fam", std::ios::out);
        if (!fam) throw Error(KGWAS_ERR_IO, "cannot create " + base + ".fam");
        for (uint64_t i = 0; i < n_acc; i++)
            fam << acc_names[i] << " " << acc_names[i] << " 0 0 0" << " " << y[i] << std::endl;
    });
}

// emma_kinship_kmers' output (src/emma_kinship_kmers.cpp:95-111)
uint64_t kgwas_kinship_format(uint64_t n_acc, const uint64_t* K, uint64_t n_used, char* out, uint64_t cap) {
    std::ostringstream os;
    for (uint64_t i = 0; i < n_acc; i++) {
        for (uint64_t j = 0; j < n_acc; j++) {
            if (j > 0) os << "\t";
            double v;
            if (i == j)
                v = 1;
            else {
                const uint64_t k = (j < i) ? K[i * n_acc + j] : K[j * n_acc + i];
                v = static_cast<double>(k) / static_cast<double>(n_used);
            }
            os << v;
        }
        os << "\n";
    }
    const std::string s = os.str();
    if (out && cap) memcpy(out, s.data(), s.size() < cap ? s.size() : cap);
    return s.size();
}

}  // extern "C"
