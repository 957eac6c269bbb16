// kmers_table_to_bed — drop-in for the reference tool of the same name (src/kmers_table_to_bed.cpp): same
// options, messages and output files (<output>.<batch>.bed/.bim/.fam); the per-row work runs on the GPU
// (kgwas_table_to_bed). Extra option: --device N.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/kgwas.h"
#include "cli_args.h"

using namespace std;

static bool file_exists(const string& fn) {
    ifstream f(fn);
    return f.good();
}
static void ck(int rc) {
    if (rc == KGWAS_OK) return;
    if (rc == KGWAS_ERR_FORMAT || rc == KGWAS_ERR_IO) {  // the reference's uncaught std::logic_error
        cerr << "terminate called after throwing an instance of 'std::logic_error'\n  what():  " << kgwas_last_error()
             << endl;
        abort();
    }
    cerr << "kmers_table_to_bed: " << kgwas_last_error() << endl;
    exit(rc == KGWAS_ERR_DEVICE ? 3 : 1);
}

int main(int argc, char* argv[]) {
    CliArgs result({
        {"kmers_table", 't', true, "k-mers table path", ""},
        {"kmers_len", 'k', true, "length of k-mers", ""},
        {"phentype_file", 'p', true, "phenotype file, condense output only to individuals with a phenotype", ""},
        {"maf", 0, true, "minor allele frequency", ""},
        {"mac", 0, true, "minor allele count", ""},
        {"batch_size", 'b', true, "maximal number of variants in each PLINK bed file (seperate to many file  if needed)", ""},
        {"output", 'o', true, "prefix for output files", ""},
        {"unique_patterns", 'u', false, "output only unique presence/absence patterns", "false"},
        {"device", 0, true, "GPU ordinal", "0"},
        {"help", 0, false, "print help", ""},
    });
    const string desc = "Convert k-mers table to PLINK binary format";
    try {
        result.parse(argc, argv);
        if (result.count("help")) {
            cerr << result.help("kmers_table_to_bed", desc) << endl;
            exit(0);
        }
        for (const char* req : {"kmers_table", "kmers_len", "phentype_file", "maf", "mac", "batch_size", "output"}) {
            if (result.count(req) == 0) {
                cerr << req << " is a required parameter" << endl;
                cerr << result.help("kmers_table_to_bed", desc) << endl;
                exit(1);
            }
        }
        const string fn_kmers_table(result.str("kmers_table"));
        const string fn_phenotypes(result.str("phentype_file"));
        const double MAF = result.f64("maf");
        const size_t MAC = result.u64("mac");
        const size_t max_batch_size = result.u64("batch_size");
        const size_t kmer_len = result.u64("kmers_len");
        const string output_base(result.str("output"));
        const bool unique_patterns = result.count("unique_patterns") > 0;
        for (const string& f : {fn_kmers_table + ".names", fn_kmers_table + ".table", fn_phenotypes}) {
            if (!file_exists(f)) {
                cerr << "Couldn't find file: " << f << endl;
                exit(1);
            }
        }
        if ((kmer_len > 31) || (kmer_len < 10)) {
            cerr << "kmer length has to be between 10-31" << endl;
            exit(1);
        }

        // phenotypes: the first column of the file; every accession must be in the table (src/kmers_table_to_bed.cpp:93-95)
        kgwas_pheno* ph = nullptr;
        ck(kgwas_pheno_load(fn_phenotypes.c_str(), &ph));
        uint64_t phenotypes_n = 0, n_accessions = 0;
        ck(kgwas_pheno_info(ph, &phenotypes_n, &n_accessions));
        if (phenotypes_n == 0) {
            cerr << "kmers_table_to_bed: phenotype file has no phenotype columns" << endl;
            exit(1);
        }
        const char* pname = nullptr;
        ck(kgwas_pheno_name(ph, 0, &pname));
        cerr << "using " << pname << endl;
        vector<const char*> acc(n_accessions);
        for (uint64_t i = 0; i < n_accessions; i++) ck(kgwas_pheno_accession(ph, i, &acc[i]));
        const float* Y = nullptr;
        ck(kgwas_pheno_values(ph, &Y));  // column 0 = the first n_accessions values

        kgwas_table* tbl = nullptr;
        ck(kgwas_table_open(fn_kmers_table.c_str(), (uint32_t)kmer_len, &tbl));
        vector<uint64_t> col(n_accessions);
        ck(kgwas_table_column_map(tbl, acc.data(), n_accessions, col.data()));

        // effective MAC (:98-100)
        size_t min_count = (size_t)ceil(double(n_accessions) * MAF);
        if (min_count < MAC) min_count = MAC;

        cerr << "loading.... " << endl;
        uint64_t n_batches = 0, n_written = 0;
        ck(kgwas_table_to_bed(tbl, col.data(), n_accessions, acc.data(), Y, min_count, max_batch_size, unique_patterns ? 1 : 0,
                              output_base.c_str(), (int)result.u64("device", 0), &n_batches, &n_written));
        for (uint64_t b = 0; b < n_batches; b++) cerr << "Batch:\t" << b + 1 << endl;
        cli_finish();
        kgwas_table_close(tbl);
        kgwas_pheno_free(ph);
    } catch (const std::invalid_argument& e) {
        cerr << "error parsing options: " << e.what() << endl;
        cerr << result.help("kmers_table_to_bed", desc) << endl;
        exit(1);
    }
    return 0;
}
