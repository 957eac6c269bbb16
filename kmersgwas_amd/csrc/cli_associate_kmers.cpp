// associate_kmers — drop-in for the reference tool of the same name (src/associate_kmers.cpp),
// same options, same output files, scoring done on the GPU through libkgwas' C ABI.
//
//   kmers_gwas.py:133-148 runs:  associate_kmers -p <pheno> -b <base> -o <dir> -n 10001 --parallel T
//        --kmers_table <table> --kmer_len 31 --maf 0.05 --mac 5 [--pattern_counter]
//        [--first_phenotype_best N]  2> log
//   outputs (src/associate_kmers.cpp:150-205): <dir>/<base>.<j>.<name>.{bed,bim,fam} per column,
//        <dir>/<base>.tested_kmers, optional <base>.<j>.best_kmers.scores.
//
// Differences from the reference, all invisible to the pipeline: the table is read once (the
// winners' rows are fetched by file row index instead of a second full pass, :167-195);
// --parallel sizes the host replay pool instead of a per-column scoring pool; extra options
// --device and --kernel select the GPU and the scoring kernel, --gpus N row-shards the table over N GPUs of the
// node inside this one process (kgwas_multiscan: contiguous shards in file order, one session and host thread per
// GPU, later shards' heap histories replayed into the first shard's heaps; output files are the same).
#include <sys/time.h>

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/kgwas.h"
#include <cstring>
#include <unistd.h>

#include "cli_args.h"

using namespace std;

static double now_s() {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec / 1e6;
}

// Data errors are uncaught std::logic_error in the reference (terminate -> SIGABRT).
[[noreturn]] static void die_logic(const string& what) {
    cerr << "terminate called after throwing an instance of 'std::logic_error'\n  what():  " << what << endl;
    abort();
}
static void ck(int rc) {
    if (rc == KGWAS_OK) return;
    if (rc == KGWAS_ERR_FORMAT || rc == KGWAS_ERR_IO) die_logic(kgwas_last_error());
    cerr << "associate_kmers: " << kgwas_last_error() << endl;
    exit(rc == KGWAS_ERR_DEVICE ? 3 : 1);
}

int main(int argc, char* argv[]) {
    const double t_main = now_s();
    CliArgs vm({
        {"phenotype_file", 'p', true, "phenotype file name", ""},
        {"base_name", 'b', true, "base name to use for all files", ""},
        {"output_dir", 'o', true, "where to save output files", "."},
        {"kmers_table", 0, true, "Presence/absemce k-mer file", ""},
        {"best", 'n', true, "Number of best k-mers to report", "1000000"},
        {"first_phenotype_best", 0, true, "if provided will save a different number of k-mers for the first phenotype", ""},
        {"batch_size", 0, true, "Loading only part of the presence absence info to memory", "10000000"},
        {"parallel", 0, true, "Max number of threads to use", "4"},
        {"kmer_len", 0, true, "Length of the k-mers", ""},
        {"maf", 0, true, "Minor allele frequency", "0.05"},
        {"mac", 0, true, "Minor allele count", "5"},
        {"k_mers_scores", 0, false, "output the best k_mers scores in binary format", ""},
        {"pattern_counter", 0, false, "Count the number of unique presence/absence patterns", ""},
        {"device", 0, true, "GPU ordinal (with --gpus: the first of N consecutive ordinals)", "0"},
        {"gpus", 0, true, "row-shard the table over this many GPUs (ordinals wrap around the GPUs present)", "1"},
        {"kernel", 0, true, "scoring kernel: 0 auto, 1 vector ALU, 2 f32 MFMA, 3 int8 coarse filter + exact re-score", "0"},
        {"help", 0, false, "print help", ""},
    });
    const string desc = "Associate k-mers presence/absence pattern with a phenotype of interest";
    try {
        vm.parse(argc, argv);
        if (vm.count("help")) {
            cerr << vm.help("associate_kmers", desc) << endl;
            exit(0);
        }
        const string fn_base = vm.str("output_dir", ".") + "/" + vm.str("base_name");
        const size_t heap_size = vm.u64("best", 1000000);
        size_t batch_size = vm.u64("batch_size", 10000000);
        const size_t threads = vm.u64("parallel", 4);
        const unsigned long long kmer_length = vm.u64("kmer_len");
        if ((kmer_length > 31) || (kmer_length < 10)) {
            cerr << "kmer length has to be between 10-31" << endl;
            exit(1);
        }
        const double maf = vm.f64("maf", 0.05);
        const size_t mac = vm.u64("mac", 5);
        const string table_base = vm.str("kmers_table");
        const string pheno_file = vm.str("phenotype_file");

        // Phenotypes (load_phenotypes_file) and the table they must all be present in.
        kgwas_pheno* ph = nullptr;
        ck(kgwas_pheno_load(pheno_file.c_str(), &ph));
        uint64_t phenotypes_n = 0, n_accessions = 0;
        ck(kgwas_pheno_info(ph, &phenotypes_n, &n_accessions));
        if (phenotypes_n == 0) die_logic("phenotype file has no phenotype columns | " + pheno_file);
        vector<const char*> acc(n_accessions), pname(phenotypes_n);
        for (uint64_t i = 0; i < n_accessions; i++) ck(kgwas_pheno_accession(ph, i, &acc[i]));
        for (uint64_t j = 0; j < phenotypes_n; j++) ck(kgwas_pheno_name(ph, j, &pname[j]));
        const float* Y = nullptr;
        ck(kgwas_pheno_values(ph, &Y));

        kgwas_table* tbl = nullptr;
        // .names is consulted first (intersect_phenotypes_to_present_DBs, :86-88), then the ctor guards
        ck(kgwas_table_open(table_base.c_str(), (uint32_t)kmer_length, &tbl));
        uint64_t S_f = 0, n_rows = 0, W_f = 0;
        ck(kgwas_table_info(tbl, &S_f, &n_rows, &W_f, nullptr));
        vector<uint64_t> col(n_accessions);
        ck(kgwas_table_column_map(tbl, acc.data(), n_accessions, col.data()));

        vector<uint64_t> topn(phenotypes_n, heap_size);
        if (vm.count("first_phenotype_best")) topn[0] = vm.u64("first_phenotype_best");

        const uint64_t min_count = kgwas_min_count(n_accessions, maf, mac);
        cerr << "Effective minor allele count:\t" << min_count << endl;

        kgwas_scan_params sp;
        sp.struct_size = sizeof(sp);
        sp.device = (int32_t)vm.u64("device", 0);
        sp.n_acc_file = S_f;
        sp.n_acc = n_accessions;
        sp.col = col.data();
        sp.n_pheno = phenotypes_n;
        sp.Y = Y;
        sp.topn = topn.data();
        sp.min_count = min_count;
        sp.chunk_rows = 0;
        // --parallel: the reference's pool of scoring tasks (src/associate_kmers.cpp:66, 104-148), 4 by default and 1 from the
        // pipeline (src/py/pipeline_parser.py:31). Here the host threads replay heap pushes beside the GPU(s). The argument is
        // HONOURED as the reference honours it - on a shared node the user's cap is the user's cap -; a value below the CPUs
        // this process may use only earns a hint on stderr, because the scan may then be host-bound. Opt-in: `--parallel 0`
        // (the reference has no use for 0) or KGWAS_AUTO_PARALLEL=1 take every CPU the process may use (cgroup quota,
        // affinity mask). Results never depend on it.
        uint64_t replay_threads = threads;
        {
            const uint64_t quota = kgwas_host_cpu_quota();
            const char* autop = opt_str("KGWAS_AUTO_PARALLEL");
            if (threads == 0 || (autop && atoi(autop) != 0)) {
                replay_threads = std::max<uint64_t>(quota, 1);
                cerr << "[kgwas] --parallel " << (threads ? "overridden by KGWAS_AUTO_PARALLEL" : "0") << ": " << replay_threads
                     << " replay threads (the CPUs this process may use)" << endl;
            } else if (replay_threads < quota) {
                cerr << "[kgwas] --parallel " << threads << (vm.count("parallel") ? "" : " (default)") << " is below the " << quota
                     << " CPUs this process may use: the replay keeps to " << threads
                     << " thread(s) and may bound the scan (--parallel 0 or KGWAS_AUTO_PARALLEL=1: all of them)" << endl;
            }
        }
        cerr << "[kgwas] replay threads requested: " << replay_threads << endl;
        sp.host_threads = (uint32_t)replay_threads;
        sp.kernel = (uint32_t)vm.u64("kernel", 0);
        sp.record_history = 0;
        sp.count_patterns = vm.count("pattern_counter") ? 1 : 0;
        // --gpus N: N contiguous row shards, shard g on GPU (device + g) modulo the GPUs present
        const uint64_t n_gpus = vm.u64("gpus", 1);
        if (n_gpus < 1 || n_gpus > 64) {
            cerr << "gpus has to be between 1-64" << endl;
            exit(1);
        }
        kgwas_scan* scan = nullptr;
        kgwas_multiscan* mscan = nullptr;
        const double t_setup = now_s();
        if (n_gpus > 1) {
            int present = 0;
            ck(kgwas_device_count(&present));
            if (present < 1) {
                cerr << "associate_kmers: no HIP device available: libkgwas has no CPU fallback" << endl;
                exit(3);
            }
            vector<int32_t> devs(n_gpus);
            for (uint64_t g = 0; g < n_gpus; g++) devs[g] = (int32_t)((sp.device + g) % (uint64_t)present);
            if ((uint64_t)present < n_gpus)
                cerr << "[kgwas] " << n_gpus << " shards on " << present << " GPU(s)" << endl;
            ck(kgwas_multiscan_create(&sp, devs.data(), (uint32_t)n_gpus, &mscan));
        } else {
            ck(kgwas_scan_create(&sp, &scan));
        }
        const double t_created = now_s();

        // Pass 1: stream the table through the GPU in file order. The reference loads a batch, then associates it
        // (src/associate_kmers.cpp:104-148); here a batch is read, copied and scored in overlapping 128 MiB pieces
        // (kgwas_scan_feed_table), so no batch-sized host buffer exists and "Load" is hidden behind "Associations".
        // The per-batch progress lines keep the reference's wording; batch_size only sets their granularity (with
        // --gpus the whole table is one batch: every GPU streams its own shard).
        if (batch_size == 0) batch_size = 1;
        if (mscan) batch_size = std::max<uint64_t>(n_rows, 1);
        double t0 = now_s(), t1;
        size_t batch_index = 0;
        for (uint64_t row0 = 0; row0 < n_rows; row0 += batch_size) {
            const uint64_t n = std::min<uint64_t>(batch_size, n_rows - row0);
            t1 = now_s();
            cerr << "Load [" << batch_index << "]\t" << (t1 - t0) / 60. << "min" << endl;
            t0 = now_s();
            if (mscan)
                ck(kgwas_multiscan_run_table(mscan, tbl, row0, n));
            else {
                if (row0 + n == n_rows) ck(kgwas_scan_expect_finish(scan));  // the last batch: finish's pops may start at its tail
                ck(kgwas_scan_feed_table(scan, tbl, row0, n));
            }
            for (uint64_t j = 0; j < phenotypes_n; j++) cerr << ".";
            t1 = now_s();
            cerr << "Associations [" << batch_index << "]\t" << (t1 - t0) / 60. << "min" << endl;
            t0 = now_s();
            batch_index++;
        }
        const double t_fed = now_s();
        kgwas_scan_stats st;
        double scan_ms = 0, merge_ms = 0;
        uint64_t rescans = 0;
        if (mscan) {
            ck(kgwas_multiscan_finish(mscan));
            ck(kgwas_multiscan_get_stats(mscan, &st, nullptr, &scan_ms, &merge_ms, &rescans));
        } else {
            ck(kgwas_scan_finish(scan));
            ck(kgwas_scan_get_stats(scan, &st));
        }

        const double t_finished = now_s();
        // Outputs (:150-205). The reference writes the winners' presence/absence from a second pass over the whole table
        // (:167-195); here the rows of ALL columns' winners are fetched by file row index and written in one batched,
        // multi-threaded call (kgwas_write_plink_many).
        vector<string> out_names(phenotypes_n);
        vector<const char*> out_bases(phenotypes_n);
        vector<uint64_t> n_win(phenotypes_n);
        vector<const uint64_t*> kmers(phenotypes_n), rows_(phenotypes_n);
        for (uint64_t j = 0; j < phenotypes_n; j++) {
            uint64_t n = 0;
            const uint64_t *kmer = nullptr, *row = nullptr;
            const double* score = nullptr;
            if (mscan)
                ck(kgwas_multiscan_result(mscan, j, &n, &kmer, &score, &row));
            else
                ck(kgwas_scan_result(scan, j, &n, &kmer, &score, &row));
            if (vm.count("k_mers_scores")) {  // output_to_file_with_scores (best_associations_heap.cpp:82-92)
                vector<char> rec(16 * n);
                for (uint64_t i = 0; i < n; i++) {
                    memcpy(&rec[16 * i], &kmer[i], sizeof(uint64_t));
                    memcpy(&rec[16 * i + 8], &score[i], sizeof(double));
                }
                ofstream of(fn_base + "." + to_string(j) + ".best_kmers.scores", ios::binary);
                of.write(rec.data(), (std::streamsize)rec.size());
            }
            out_names[j] = fn_base + "." + to_string(j) + "." + pname[j];
            out_bases[j] = out_names[j].c_str();
            n_win[j] = n;
            kmers[j] = kmer;
            rows_[j] = row;
            cerr << "Save [" << j << "]" << endl;
        }
        ck(kgwas_write_plink_many(phenotypes_n, out_bases.data(), tbl, col.data(), n_accessions, acc.data(), Y, n_win.data(), kmers.data(),
                                  rows_.data(), (uint32_t)replay_threads));
        const double t_written = now_s();
        if (vm.count("pattern_counter")) {  // :143-144, :197-201
            cerr << "Total patterns\t" << st.patterns << endl;
            ofstream fout(fn_base + ".pattern_counter");
            fout << st.patterns << endl;
        }
        {
            ofstream fout(fn_base + ".tested_kmers");
            fout << st.rows_tested << endl;
        }
        cerr << "[kgwas] kernel="
             << (st.kernel_used == KGWAS_KERNEL_COARSE   ? "coarse_filter+exact"
                 : st.kernel_used == KGWAS_KERNEL_NARROW ? "narrow_fp4+exact"
                 : st.kernel_used == KGWAS_KERNEL_MFMA   ? "mfma_f32"
                                                         : "valu")
             << " direct=" << st.direct_mode << " chunks=" << st.chunks << " score_kernel_ms=" << st.score_kernel_ms
             << " candidates=" << st.candidates << " heap_pushes=" << st.heap_pushes << endl;
        cerr << "[kgwas] replay_threads=" << replay_threads << " replay_threads_per_gpu=" << std::max<uint64_t>(1, replay_threads / n_gpus) << endl;
        if (mscan)
            cerr << "[kgwas] gpus=" << n_gpus << " scan_ms=" << scan_ms << " merge_ms=" << merge_ms << " rescans=" << rescans << endl;
        // where the wall time of the run went (bench.py's cli_e2e record reads this line)
        cerr << "[kgwas] seconds: setup=" << (t_setup - t_main) << " session_create=" << (t_created - t_setup) << " scan=" << (t_fed - t_created)
             << " finish=" << (t_finished - t_fed) << " output=" << (t_written - t_finished) << " total=" << (now_s() - t_main) << endl;
        const double t_down = now_s();
        cli_finish();  // (returns only with KGWAS_CLI_FULL_TEARDOWN=1)
        if (mscan) kgwas_multiscan_destroy(mscan);
        if (scan) kgwas_scan_destroy(scan);
        kgwas_table_close(tbl);
        kgwas_pheno_free(ph);
        cerr << "[kgwas] teardown_s=" << (now_s() - t_down) << endl;
    } catch (const std::invalid_argument& e) {
        cerr << "error parsing options: " << e.what() << endl;
        cerr << vm.help("associate_kmers", desc) << endl;
        exit(1);
    }
    return 0;
}
