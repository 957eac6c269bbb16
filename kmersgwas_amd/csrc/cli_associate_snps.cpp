// associate_snps — drop-in for the reference tool of the same name (src/associate_snps.cpp): six positional
// arguments, same stderr lines, <base output>.<phenotype>.bed/.bim with the most associated SNPs of every phenotype
// column; scoring runs on the GPU (kgwas_snps_*). KGWAS_DEVICE selects the GPU ordinal.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "cli_args.h"
#include "../../include/kgwas.h"

using namespace std;

static void ck(int rc) {
    if (rc == KGWAS_OK) return;
    if (rc == KGWAS_ERR_FORMAT || rc == KGWAS_ERR_IO) {  // the reference's uncaught std::logic_error
        cerr << "terminate called after throwing an instance of 'std::logic_error'\n  what():  " << kgwas_last_error()
             << endl;
        abort();
    }
    cerr << "associate_snps: " << kgwas_last_error() << endl;
    exit(rc == KGWAS_ERR_DEVICE ? 3 : 1);
}

int main(int argc, char* argv[]) {
    if (argc != 7) {
        cerr << "usage: " << argv[0]
             << " <phenotypes file> <base bedbim file> <base output files> <# snps to output> <maf> <mac>" << endl;
        return 1;
    }
    // load phenotypes
    kgwas_pheno* ph = nullptr;
    ck(kgwas_pheno_load(argv[1], &ph));
    uint64_t phenotype_n = 0, n_samples = 0;
    ck(kgwas_pheno_info(ph, &phenotype_n, &n_samples));
    vector<const char*> acc(n_samples), pname(phenotype_n);
    for (uint64_t i = 0; i < n_samples; i++) ck(kgwas_pheno_accession(ph, i, &acc[i]));
    for (uint64_t j = 0; j < phenotype_n; j++) ck(kgwas_pheno_name(ph, j, &pname[j]));
    const float* Y = nullptr;
    ck(kgwas_pheno_values(ph, &Y));
    cerr << "Loading snps information" << endl;
    kgwas_snps* snps = nullptr;
    ck(kgwas_snps_open(argv[2], acc.data(), n_samples, &snps));

    const size_t n_best_snps_to_save = (size_t)atoi(argv[4]);
    const double maf = atof(argv[5]);  // minor allele frequency
    cerr << "MAF = " << maf << " n_sample = " << n_samples << endl;
    double mac = atof(argv[6]);
    if (mac < ceil(maf * n_samples)) mac = ceil(maf * n_samples);  // minor allele count
    cerr << "Minor allele count  = " << mac << endl;
    const int device = (int)opt_int("KGWAS_DEVICE", 0);
    cerr << "Associating phenotypes:";
    cerr.flush();
    const auto t0 = chrono::steady_clock::now();
    vector<uint64_t> counts(phenotype_n), indices(phenotype_n * max<size_t>(n_best_snps_to_save, 1));
    ck(kgwas_snps_best(snps, Y, phenotype_n, n_best_snps_to_save, mac, device, counts.data(), indices.data()));
    for (uint64_t j = 0; j < phenotype_n; j++) cerr << ".";
    const double d_time = chrono::duration<double>(chrono::steady_clock::now() - t0).count() / (double)max<uint64_t>(phenotype_n, 1);
    cerr << "Average time per phenotype:\t" << d_time << endl;
    cerr << "\noutputting best snps";
    vector<string> bases;
    for (uint64_t j = 0; j < phenotype_n; j++) bases.push_back(string(argv[3]) + "." + pname[j]);
    vector<const char*> bases_c;
    for (auto& b : bases) bases_c.push_back(b.c_str());
    ck(kgwas_snps_write(snps, phenotype_n, bases_c.data(), counts.data(), indices.data(), n_best_snps_to_save));
    cli_finish();
    kgwas_snps_close(snps);
    kgwas_pheno_free(ph);
    return 0;
}
