// emma_kinship_kmers — drop-in for the reference tool of the same name
// (src/emma_kinship_kmers.cpp): same options (-t/--kmers_table, -k/--kmers_len, --maf, all
// required), matrix on stdout, progress on stderr; the accumulation runs on the GPU.
#include <chrono>
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
    if (rc == KGWAS_ERR_FORMAT || rc == KGWAS_ERR_IO) {
        cerr << "terminate called after throwing an instance of 'std::logic_error'\n  what():  " << kgwas_last_error()
             << endl;
        abort();
    }
    cerr << "emma_kinship_kmers: " << kgwas_last_error() << endl;
    exit(rc == KGWAS_ERR_DEVICE ? 3 : 1);
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char* argv[]) {
    CliArgs result({
        {"kmers_table", 't', true, "k-mers table path", ""},
        {"kmers_len", 'k', true, "length of k-mers", ""},
        {"maf", 0, true, "minor allele frequency", ""},
        {"device", 0, true, "GPU ordinal (with --gpus: the first of N consecutive ordinals)", "0"},
        {"gpus", 0, true, "row-shard the table over this many GPUs (ordinals wrap around the GPUs present)", "1"},
        {"help", 0, false, "print help", ""},
    });
    const string desc = "Calculate a kinship matrix from the k-mers table (output to stdout)";
    const double t_main = now_s();
    try {
        result.parse(argc, argv);
        if (result.count("help")) {
            cerr << result.help("emma_kinship_kmers", desc) << endl;
            exit(0);
        }
        for (const char* req : {"kmers_table", "kmers_len", "maf"}) {
            if (result.count(req) == 0) {
                cerr << req << " is a required parameter" << endl;
                cerr << result.help("emma_kinship_kmers", desc) << endl;
                exit(1);
            }
        }
        const string fn_kmers_table(result.str("kmers_table"));
        const double MAF = result.f64("maf");
        const size_t kmer_len = result.u64("kmers_len");
        for (const string& f : {fn_kmers_table + ".names", fn_kmers_table + ".table"}) {
            if (!file_exists(f)) {
                cerr << "Couldn't find file: " << f << endl;
                exit(1);
            }
        }
        if ((kmer_len > 31) || (kmer_len < 10)) {
            cerr << "kmer length has to be between 10-31" << endl;
            exit(1);
        }
        kgwas_table* tbl = nullptr;
        ck(kgwas_table_open(fn_kmers_table.c_str(), (uint32_t)kmer_len, &tbl));
        uint64_t n_acc = 0, n_rows = 0, W_f = 0;
        ck(kgwas_table_info(tbl, &n_acc, &n_rows, &W_f, nullptr));
        const size_t min_count = (size_t)ceil(static_cast<double>(n_acc) * MAF);  // :83
        cerr << "Min count = " << min_count << endl;
        const uint64_t n_gpus = result.u64("gpus", 1);
        if (n_gpus < 1 || n_gpus > 64) {
            cerr << "gpus has to be between 1-64" << endl;
            exit(1);
        }
        vector<uint64_t> H(n_acc * n_acc), K(n_acc * n_acc);
        uint64_t n_snps = 0;
        const double t_setup = now_s();
        double t_created = t_setup;
        cerr << "loading..." << endl;
        if (n_gpus > 1) {
            // contiguous row shards, one session + thread per GPU, integer partials added (kgwas_kinship_table_multi)
            int present = 0;
            ck(kgwas_device_count(&present));
            if (present < 1) {
                cerr << "emma_kinship_kmers: no HIP device available: libkgwas has no CPU fallback" << endl;
                exit(3);
            }
            vector<int32_t> devs(n_gpus);
            for (uint64_t g = 0; g < n_gpus; g++) devs[g] = (int32_t)((result.u64("device", 0) + g) % (uint64_t)present);
            ck(kgwas_kinship_table_multi(devs.data(), (uint32_t)n_gpus, tbl, min_count, H.data(), &n_snps));
        } else {
            kgwas_kinship* kin = nullptr;
            ck(kgwas_kinship_create((int32_t)result.u64("device", 0), n_acc, min_count, &kin));
            t_created = now_s();
            // the reference loads 2^20 rows, then accumulates them and prints a dot (:89-93); here the file read, the copy and
            // the kernels of consecutive pieces overlap (kgwas_kinship_feed_table), in feeds of 2^26 rows (a feed ends with
            // the pipeline drained), and the dots - one per 2^20 rows, as there - follow each feed
            const uint64_t batch = 1ull << 26;
            for (uint64_t row0 = 0; row0 < n_rows; row0 += batch) {
                const uint64_t n = std::min<uint64_t>(batch, n_rows - row0);
                ck(kgwas_kinship_feed_table(kin, tbl, row0, n));
                for (uint64_t d = 0; d < (n + (1ull << 20) - 1) >> 20; d++) cerr << ".";
                cerr.flush();
            }
            ck(kgwas_kinship_partials(kin, H.data(), &n_snps));
            if (opt_str("KGWAS_CLI_FULL_TEARDOWN")) kgwas_kinship_destroy(kin);  // (otherwise left to the process exit, cli_finish)
        }
        const double t_fed = now_s();
        ck(kgwas_kinship_from_partials(n_acc, H.data(), n_snps, K.data()));
        cerr << "#" << n_snps << endl;
        // (one call: a cell is at most 12 characters - six significant digits, a point or an exponent - and a separator)
        string text(n_acc * n_acc * 16 + 16, '\0');
        const uint64_t need = kgwas_kinship_format(n_acc, K.data(), n_snps, &text[0], text.size());
        if (need > text.size()) {
            text.assign(need, '\0');
            kgwas_kinship_format(n_acc, K.data(), n_snps, &text[0], need);
        } else
            text.resize(need);
        const double t_text = now_s();
        cout << text;
        cout.flush();
        // where the wall time of the run went (as associate_kmers' last line)
        cerr << "[kgwas] seconds: setup=" << (t_setup - t_main) << " session_create=" << (t_created - t_setup) << " accumulate=" << (t_fed - t_created)
             << " matrix_text=" << (t_text - t_fed) << " stdout=" << (now_s() - t_text) << " total=" << (now_s() - t_main) << endl;
        cli_finish();
        kgwas_table_close(tbl);
    } catch (const std::invalid_argument& e) {
        cerr << "error parsing options: " << e.what() << endl;
        cerr << result.help("emma_kinship_kmers", desc) << endl;
        exit(1);
    }
    return 0;
}
