// scan_create.cpp — kgwas_scan_create: validation, the constant device data of a session (phenotype layouts of the exact
// scorers, the operand sets and error bounds of the filters: block-scaled FP4 x FP6/FP4, int8, narrow), buffers and slots.
#include "scan_internal.h"
#include <chrono>

extern "C" {

int kgwas_scan_create(const kgwas_scan_params* p, kgwas_scan** out) {
    return guarded([&] {
        if (!p || !out) throw Error(KGWAS_ERR_ARG, "kgwas_scan_create: null argument");
        if (p->struct_size != sizeof(kgwas_scan_params)) throw Error(KGWAS_ERR_ARG, "kgwas_scan_params: size mismatch");
        if (!p->col || !p->Y || !p->topn || p->n_acc == 0 || p->n_pheno == 0 || p->n_acc_file == 0)
            throw Error(KGWAS_ERR_ARG, "kgwas_scan_create: empty problem");
        if (p->n_acc > p->n_acc_file) throw Error(KGWAS_ERR_ARG, "more phenotyped accessions than table columns");
        if (p->n_acc_file >= (1ull << 31)) throw Error(KGWAS_ERR_ARG, "too many accessions");
        require_heap_emulation();  // tie order = libstdc++'s heap moves (src/kmer_general.h:113-128): verified against THIS process's library
        check_device(p->device);
        KGWAS_HIP(hipSetDevice(p->device));
        std::unique_ptr<kgwas_scan> s(new kgwas_scan);
        s->device = p->device;
        s->S_f = p->n_acc_file;
        s->S = p->n_acc;
        s->W_f = (s->S_f + 63) / 64;
        s->W_m = 2 * ((s->S + 127) / 128);  // src/kmers_multiple_databases.cpp:51
        s->L = 64 * s->W_m;
        s->n_pheno = p->n_pheno;
        s->min_count = p->min_count;
        s->col.assign(p->col, p->col + s->S);
        s->topn.assign(p->topn, p->topn + s->n_pheno);
        s->Y.assign(p->Y, p->Y + s->n_pheno * s->S);
        if (p->record_history > 2) throw Error(KGWAS_ERR_ARG, "record_history: 0 (off), 1 (full log) or 2 (eviction ring)");
        s->record_history = p->record_history == 1;
        if (p->record_history == 2) {
            s->history_ring = 1;  // per heap: 16 standard deviations of the rank distance between two shards' N-th scores
            if (const char* e = opt_str("KGWAS_HISTORY_RING"))
                if (atoll(e) > 0) s->history_ring = (size_t)atoll(e);
        }
        s->count_patterns = p->count_patterns != 0;
        std::vector<bool> seen(s->S_f, false);
        for (uint64_t i = 0; i < s->S; i++) {
            if (s->col[i] >= s->S_f) throw Error(KGWAS_ERR_ARG, "column index out of range");
            if (seen[s->col[i]]) throw Error(KGWAS_ERR_ARG, "duplicate column index");
            seen[s->col[i]] = true;
        }
        for (uint64_t j = 0; j < s->n_pheno; j++) {
            if (s->topn[j] == 0) throw Error(KGWAS_ERR_ARG, "heap size must be >= 1");
            s->max_topn = std::max(s->max_topn, s->topn[j]);
            s->sum_topn += s->topn[j];
        }
        s->direct = true;
        for (uint64_t i = 0; i < s->S; i++) s->direct = s->direct && (s->col[i] == i);

        bool finite = true;
        for (float v : s->Y) finite = finite && std::isfinite(v);
        // The filters bound |reference yigi - exact sum| by float32 ROUNDING errors only: that needs every partial sum of
        // the reference's four chains (and their final adds) to stay finite. sum |y_i| of a column, with the growth factor
        // of recursive float32 summation, bounds them all; a column beyond that (|y| ~ 1e36 and up) can overflow to +-inf
        // in the reference on rows whose exact sum is finite - such sessions keep the exact scorers, which reproduce the
        // overflow (tests/test_gpu_parity.py::test_numeric_edges_of_the_phenotype_values[huge]).
        bool chain_safe = finite;
        for (uint64_t j = 0; j < s->n_pheno && chain_safe; j++) {
            double a = 0;
            for (uint64_t i = 0; i < s->S; i++) a += std::fabs((double)s->Y[j * s->S + i]);
            chain_safe = a * (1.0 + (double)(s->S + 8) * 0x1p-23) < 0.99 * (double)std::numeric_limits<float>::max();
        }
        uint32_t kern = p->kernel;
        const bool mfma_fits = mfma_lds_bytes((uint32_t)s->W_m) <= 160u * 1024u;
        // Coarse int8 filter + exact re-scoring for the sparse phase (score_coarse.hip): needs finite values,
        // an exact kernel for the dense phase / re-runs, and T >= 2 int8 tiles of the whole sample axis in LDS.
        const uint32_t n_kgroups = (uint32_t)((s->W_m + 7) / 8);
        uint32_t coarse_T = 0;
        for (uint32_t T : {8u, 7u, 6u, 5u, 4u, 3u, 2u})
            if (coarse_lds_bytes(n_kgroups, T) <= 152u * 1024u) {
                coarse_T = T;
                break;
            }
        // The operand-streaming form of the block-scaled filter (score_mxs.hip) holds one step's operands in LDS, not a whole
        // column tile's: no limit on the accessions. KGWAS_MXS: 0 never; 1 (default) where the resident form does not exist
        // (no column tile's operands fit the LDS) or would pass every row through several LDS groups of ONE or TWO column
        // tiles (its matrix instructions run at a fraction of a full group's efficiency there); 2 wherever the resident form
        // needs more than one LDS group - measured level with it, not ahead: 2048 x 201 40.4-41.0 against 39.2-40.3 ms per
        // 100 M rows, 1135 x 101 13.3 against 13.4 -; 3 wherever the form exists. The int8 filter
        // (KGWAS_COARSE_MX=0) stops at 5120 accessions.
        const int mxs_want = (int)opt_int("KGWAS_MXS", 1);
        const bool mxs_can = mxs_want != 0 && !(opt_int("KGWAS_COARSE_MX", 1) == 0) && !opt_set("KGWAS_COARSE_SLICES") &&
                             !(opt_int("KGWAS_MX_S1", -1) == 6);
        const bool filter_fits = coarse_T != 0 || mxs_can;
        bool want_coarse = false;
        if (kern == KGWAS_KERNEL_COARSE) {
            if (!chain_safe || !filter_fits)
                throw Error(KGWAS_ERR_ARG, "coarse filter needs finite phenotype values whose float32 sums cannot overflow (sum |y| < FLT_MAX per column) and, for its int8 form, <= 5120 accessions");
            want_coarse = true;
            kern = KGWAS_KERNEL_AUTO;
        } else if (kern == KGWAS_KERNEL_AUTO && chain_safe && filter_fits) {
            // any number of columns: even a single column (one mostly empty 16-column tile) runs twice as fast behind
            // the filter as through the exact VALU scorer (12.5 vs 27 ms per 100 M-row pass)
            want_coarse = true;
        }
        if (kern == KGWAS_KERNEL_AUTO) kern = (s->n_pheno >= 4 && finite && mfma_fits) ? KGWAS_KERNEL_MFMA : KGWAS_KERNEL_VALU;
        if (kern == KGWAS_KERNEL_MFMA && !mfma_fits)
            throw Error(KGWAS_ERR_ARG, "MFMA scorer: phenotype tile does not fit LDS for this many accessions");
        if (kern == KGWAS_KERNEL_MFMA && !finite)
            throw Error(KGWAS_ERR_ARG, "MFMA scorer needs finite phenotype values (0*inf); use the VALU scorer");
        if (kern != KGWAS_KERNEL_MFMA && kern != KGWAS_KERNEL_VALU) throw Error(KGWAS_ERR_ARG, "unknown kernel id");
        s->kernel_used = kern;
        s->coarse = want_coarse;
        s->coarse_T = coarse_T;
        s->n_kgroups = n_kgroups;
        // One to four columns under AUTO: the narrow filter (FP4 x FP8 block-scaled MFMA, three slices per column)
        // instead of the int8 one, whose 16-column tiles would be mostly padding (KGWAS_NARROW=0: keep the int8 filter).
        s->narrow = want_coarse && p->kernel == KGWAS_KERNEL_AUTO && s->n_pheno <= NARROW_MAX_COLS &&
                    narrow_lds_bytes(n_kgroups) <= 64u * 1024u && !(opt_int("KGWAS_NARROW", 1) == 0);

        // (narrow filter on rows read in place: chunks of up to 128 M rows - with one column a chunk's fixed costs, five
        // launches and a copy with the gaps between them, ~40 us, weigh more than the candidates a staler threshold lets
        // through. 1.2 G rows x 1024 samples, one column: cap 32 M rows 49 chunks 30.8 ms, 64 M 33 / 30.0, 128 M 25 / 29.8,
        // 256 M 21 / 29.7 - identical heaps, tools/p1_large_chunks.py)
        s->chunk_max = p->chunk_rows ? p->chunk_rows : ((s->narrow && s->direct) ? (128ull << 20) : (8ull << 20));
        s->chunk_max = std::max<uint64_t>(128, (s->chunk_max + 127) / 128 * 128);
        // test hook (kgwas_scan_debug_residuals): keep every filter form's quantisation residuals, so that a test can build the
        // rows on which the bound |yigi_ref - yc| <= Eg + min(Rall, N1 * rmax) is TIGHT (tests/test_gpu_parity.py, adversarial bound)
        if (s->coarse && opt_str("KGWAS_DEBUG_RESIDUALS")) {
            s->dbg_keep_resid = true;
            for (auto& v : s->dbg_resid) v.assign(s->n_pheno * s->S, 0.0);
        }
        if (s->coarse) {  // survivor keys are (column << row_bits | row) in 32 bits, the 0xFFFFFFFF fill included
            uint32_t pbits = 1;
            while ((1ull << pbits) < s->n_pheno + 1) pbits++;
            if (pbits > 22) throw Error(KGWAS_ERR_ARG, "coarse filter: too many phenotype columns for 32-bit survivor keys");
            s->chunk_max = std::max<uint64_t>(128, std::min<uint64_t>(s->chunk_max, 1ull << (32 - pbits)));
        }
        if (s->coarse && !s->narrow) {  // the coarse kernel addresses a chunk's rows with 32-bit byte offsets
            const uint64_t stride_dw = 2 * (1 + std::max<uint64_t>(s->W_f, s->W_m));
            const uint64_t lim = ((1ull << 32) - (1ull << 20)) / (4 * stride_dw) / 128 * 128;
            s->chunk_max = std::max<uint64_t>(128, std::min<uint64_t>(s->chunk_max, lim));
        }
        s->dense_rows = std::min<uint64_t>(16384, s->chunk_max);
        // Dense chunks of a feed: enough rows to fill the largest heap with a margin for the MAC filter (more
        // dense chunks follow while a heap is still short); everything after goes through the sparse path.
        s->dense_chunk = std::min<uint64_t>(s->dense_rows, std::max<uint64_t>(1024, (s->max_topn + s->max_topn / 8 + 512 + 127) / 128 * 128));
        if (exp_str("KGWAS_MODE_K")) s->mode_k = atof(exp_str("KGWAS_MODE_K"));  // experiments
        const uint64_t budget = (uint64_t)exp_int("KGWAS_CAP_BUDGET", (long long)((4ull << 20)));  // candidate records per slot
        // (few columns: longer lists, so that the ramp takes ~6 chunks instead of ~13 - a chunk's fixed costs, not its
        // rows, are what a one-column scan pays for)
        const uint64_t cap_mult = (uint64_t)exp_int("KGWAS_CAP_MULT", (long long)((s->narrow ? 16 : 2)));  // experiments
        uint64_t cap = std::min<uint64_t>(cap_mult * s->max_topn + 4096, std::max<uint64_t>(budget / s->n_pheno, 1024));
        s->cap = (uint32_t)std::min<uint64_t>(cap, 0x7FFFFFFFull);

        const bool trace_create = opt_set("KGWAS_TRACE");
        const auto tc0 = std::chrono::steady_clock::now();
        auto tcreate = [&](const char* what) {
            if (trace_create)
                fprintf(stderr, "[kgwas] scan_create +%.1f ms: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count(), what);
        };
        KGWAS_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        tcreate("stream created (HIP context up)");
        // (Round 4 tried a side thread here that loaded the library's code objects and exercised the stream's first launch, copy
        // and cross-stream wait while this thread went on: ~20 ms of a fresh process. A fuzz run then hung INSIDE this function -
        // one helper thread spinning, this thread blocked - about once per 3 500 s of randomised sessions, never before that
        // thread existed: two threads driving the HIP runtime's allocation / registration / synchronisation paths at once is
        // not worth 20 ms. The first uses are paid where they occur.)
        KGWAS_HIP(hipEventCreate(&s->ev_user));
        KGWAS_HIP(hipEventCreate(&s->ev_ds));
        KGWAS_HIP(hipEventCreateWithFlags(&s->ev_dcopy, hipEventDisableTiming));
        KGWAS_HIP(hipEventCreate(&s->ev_d0));
        KGWAS_HIP(hipEventCreate(&s->ev_d1));

        tcreate("events created");
        // ---- constant device data --------------------------------------------------------
        const uint64_t S = s->S, L = s->L, W_m = s->W_m, P = s->n_pheno;
        std::vector<uint32_t> dmask(2 * W_m, 0), colmap(L, 0xFFFFFFFFu);
        for (uint64_t d = 0; d < 2 * W_m; d++) {
            if (!s->direct)
                dmask[d] = 0xFFFFFFFFu;
            else if (32 * d + 32 <= S)
                dmask[d] = 0xFFFFFFFFu;
            else if (32 * d < S)
                dmask[d] = (1u << (S - 32 * d)) - 1u;
        }
        for (uint64_t i = 0; i < S; i++) colmap[i] = (uint32_t)s->col[i];
        const uint64_t P4 = (P + 3) / 4 * 4, nct = (P + 15) / 16;
        std::vector<float> Yperm(P4 * L, 0.0f), Ymfma(nct * L * 16, 0.0f), sums(P, 0.0f);
        std::vector<float> V(L);
        for (uint64_t j = 0; j < P; j++) {
            std::fill(V.begin(), V.end(), 0.0f);
            for (uint64_t i = 0; i < S; i++) V[i] = s->Y[j * S + i];
            float* R = &Yperm[j * L];
            // permute_scores (src/kmer_general.cpp:155-167): R[128b+4s+l] = V[128b+32l+31-s]
            for (uint64_t b = 0; b < L / 128; b++)
                for (uint64_t sx = 0; sx < 32; sx++)
                    for (uint64_t l = 0; l < 4; l++) R[128 * b + 4 * sx + l] = V[128 * b + 32 * l + 31 - sx];
            // update_scores_and_sum (src/kmers_multiple_databases.cpp:288-295): sequential float32 sum
            volatile float sum = 0.0f;
            for (uint64_t i = 0; i < L; i++) sum = sum + R[i];
            sums[j] = sum;
            // MFMA layout (see score_mfma.hip): [ct][(((b*4+l)*2 + t/4)*64 + kk*16+n)*4 + t%4], s = 4t+kk
            const uint64_t ct = j / 16, n = j % 16;
            for (uint64_t b = 0; b < L / 128; b++)
                for (uint64_t l = 0; l < 4; l++)
                    for (uint64_t sx = 0; sx < 32; sx++)
                    {
                        // chain step sx = 4t + kk; lane = kk*16 + n; four consecutive t sit together
                        const uint64_t t = sx / 4, kk = sx % 4;
                        Ymfma[ct * L * 16 + (((b * 4 + l) * 2 + t / 4) * 64 + kk * 16 + n) * 4 + t % 4] =
                            V[128 * b + 32 * l + 31 - sx];
                    }
        }
        {
            const uint64_t avail = s->direct ? 2 * s->W_f : 2 * W_m;
            uint32_t nb = 0;
            while (nb < W_m / 2 && 4ull * nb + 3 < avail && dmask[4 * nb] == 0xFFFFFFFFu &&
                   dmask[4 * nb + 1] == 0xFFFFFFFFu && dmask[4 * nb + 2] == 0xFFFFFFFFu && dmask[4 * nb + 3] == 0xFFFFFFFFu)
                nb++;
            s->nb_full = nb;
        }
        tcreate("phenotype layouts built (host)");
        s->d_dmask.alloc(dmask.size());
        s->d_colmap.alloc(colmap.size());
        s->d_sums.alloc(P);
        s->d_thr.alloc(P);
        s->h_thr.alloc(8 * P);
        s->d_thr_host.alloc(P);
        s->d_thr_redo.alloc(P);
        s->h_thr_redo.alloc(P);
        s->d_hist.alloc(P * (size_t)HIST_BINS);
        s->d_hist_base.alloc(P);
        s->h_hist_base.alloc(P);
        s->d_pat_cnt.alloc(1);
        KGWAS_HIP(hipMemset(s->d_pat_cnt.p, 0, 8));
        KGWAS_HIP(hipMemset(s->d_thr.p, 0, P * sizeof(double)));  // 0 = "nothing is filtered" until the heaps say otherwise
        KGWAS_HIP(hipMemset(s->d_thr_host.p, 0, P * sizeof(double)));
        s->d_topn.alloc(P);
        s->d_sel.alloc(P);
        s->h_sel.alloc(P);
        s->d_sel_info.alloc(2);
        s->h_sel_info.alloc(2);
        KGWAS_HIP(hipMemcpy(s->d_topn.p, s->topn.data(), P * 8, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(s->d_dmask.p, dmask.data(), dmask.size() * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(s->d_colmap.p, colmap.data(), colmap.size() * 4, hipMemcpyHostToDevice));
        KGWAS_HIP(hipMemcpy(s->d_sums.p, sums.data(), P * 4, hipMemcpyHostToDevice));
        if (kern == KGWAS_KERNEL_MFMA) {
            s->d_Ymfma.alloc(Ymfma.size());
            KGWAS_HIP(hipMemcpy(s->d_Ymfma.p, Ymfma.data(), Ymfma.size() * 4, hipMemcpyHostToDevice));
        }
        if (kern != KGWAS_KERNEL_MFMA || s->coarse) {
            s->d_Yperm.alloc(Yperm.size());
            KGWAS_HIP(hipMemcpy(s->d_Yperm.p, Yperm.data(), Yperm.size() * 4, hipMemcpyHostToDevice));
        }
        tcreate("small device buffers allocated and uploaded");
        if (s->coarse) {
            // int8 slices per column: y_i ~ c + u*(254*q0_i + q1_i) (two slices, ~15 bits) or c + u*q0_i (one),
            // centred at c = sum/N, sum being the reference's float32 sum of the column: then
            //   r_c = N*yc - N1*sum = N*u*Dc + N1*(N*c - sum),   |N1*(N*c - sum)| <= rho  (rounding of c only),
            // i.e. an exact integer Dc times a constant. For every row
            //   |yigi_ref - yc| <= Eg + |sum_{i in row} resid_i| <= Eg + min(Rall, N1 * rmax):
            //   Eg   = gamma_{L/4+3} * sum|y_i|  float32 summation error of the reference chains (Higham, recursive sums)
            //   Rall = max(sum of the positive resid_i, sum of the |negative resid_i|)  (a row's residuals cannot
            //          add up to more than all residuals of one sign), rmax = max_i |resid_i|,
            //          resid_i = y_i - c - u*(254 q0_i + q1_i)
            // so score_ref > thr needs (N*u*|Dc| + rho + N*E)^2 >= thr*d*(1 - 2^-40), i.e.
            //   |Dc| >= sqrt(thr)*kalpha*sqrt(d) - eg - min(rall, N1*rmax)       (units of u; score_coarse.hip)
            // with kalpha rounded down by 2^-19 relative and the error terms rounded up and padded: the device
            // evaluates the right-hand side in float32, and these margins dominate its rounding.
            const double u32 = std::ldexp(1.0, -24);
            const double nterms = (double)L / 4.0 + 3.0;
            const double gamma = nterms * u32 / (1.0 - nterms * u32);
            std::vector<int> q0(S), q1(S);
            auto up = [](double x) { return std::nextafter((float)x, std::numeric_limits<float>::infinity()); };
            struct ErrBound {
                float eg, rall, rmax;     // phenotype units, rounded up
                float egD, rallD, rmaxD;  // the same in units of Dc (divided by u), rounded up: what the kernel uses
            };
            auto quantise = [&](uint64_t j, int ns, CoarseCol& cc, ErrBound& eb) {
                const double Nd = (double)S, sum = (double)sums[j];
                const double c = sum / Nd;
                double mx = 0, A = 0;
                for (uint64_t i = 0; i < S; i++) {
                    const double y = (double)s->Y[j * S + i];
                    mx = std::max(mx, std::fabs(y - c));
                    A += std::fabs(y);
                }
                // unit u: one slice spans +-127 u, two slices +-(127*254 + 127) u
                const double u = mx > 0 ? (ns == 2 ? mx / (127.0 * 254.0) : mx / 127.0) : 1.0;
                const double a0 = ns == 2 ? 254.0 * u : u;
                double rpos = 0, rneg = 0, rmax = 0;
                for (uint64_t i = 0; i < S; i++) {
                    const double y = (double)s->Y[j * S + i] - c;
                    int v0 = (int)std::lrint(y / a0);
                    v0 = std::max(-127, std::min(127, v0));
                    double r = y - a0 * v0;
                    int v1 = 0;
                    if (ns == 2) {
                        v1 = (int)std::lrint(r / u);
                        v1 = std::max(-127, std::min(127, v1));
                        r -= u * v1;
                    }
                    q0[i] = v0;
                    q1[i] = v1;
                    if (s->dbg_keep_resid) s->dbg_resid[ns - 1][j * S + i] = r;
                    if (r > 0) rpos += r; else rneg -= r;
                    rmax = std::max(rmax, std::fabs(r));
                }
                const double rho = Nd * std::fabs(Nd * c - sum) * 2.0 + 1e-9 * (1.0 + std::fabs(sum));
                const double Eg = gamma * A * (1.0 + 1e-6) + 1e-12 * (1.0 + A);
                cc.kalpha = (1.0 - std::ldexp(1.0, -19)) / (Nd * u);
                cc.iu = up(1.0 / u * (1.0 + 1e-6));
                eb.eg = up((Eg + rho / Nd) * (1.0 + 1e-6) + 1e-30);
                eb.rall = up(std::max(rpos, rneg) * (1.0 + 1e-6));
                eb.rmax = up(rmax * (1.0 + 1e-6));
                const double iu = 1.0 / u * (1.0 + 1e-6);
                eb.egD = up((double)eb.eg * iu);
                eb.rallD = up((double)eb.rall * iu);
                eb.rmaxD = up((double)eb.rmax * iu);
            };
            // One slice halves the matrix work but widens the bound; it is offered when, for every column, the bound
            // at N1 = S/2 stays below 15 % of the deviation of yigi a z = 4 association needs (2*sigma*sqrt(S)), so the
            // survivors stay within a small multiple of the true candidates. KGWAS_COARSE_SLICES=1|2 forces one set.
            bool one_ok = true;
            for (uint64_t j = 0; j < P && one_ok; j++) {
                CoarseCol cc;
                ErrBound eb;
                quantise(j, 1, cc, eb);
                double mean = 0, var = 0;
                for (uint64_t i = 0; i < S; i++) mean += (double)s->Y[j * S + i];
                mean /= (double)S;
                for (uint64_t i = 0; i < S; i++) var += ((double)s->Y[j * S + i] - mean) * ((double)s->Y[j * S + i] - mean);
                const double sigma = std::sqrt(var / (double)S);
                const double e_half = (double)eb.eg + std::min((double)eb.rall, 0.5 * (double)S * (double)eb.rmax);
                if (!(e_half <= 0.15 * 2.0 * sigma * std::sqrt((double)S))) one_ok = false;
            }
            bool want[2] = {one_ok, true};
            if (const char* e = opt_str("KGWAS_COARSE_SLICES")) {
                if (atoi(e) == 1) want[0] = true, want[1] = false;
                if (atoi(e) == 2) want[0] = false, want[1] = true;
            }
            if (s->narrow) {
                want[0] = want[1] = false;  // the int8 operand sets are not needed
                // FP8 E4M3 operands of the narrow filter (score_narrow.hip): three slices of integers in [-15, 15] per
                // column, y_i - c ~ sum_k u_k q_ki with u_0 = max|y_i - c| / 15 and u_{k+1} = u_k / 30 (a rounding
                // residual of at most u_k / 2 fills the next slice's range exactly), and a ones row per column.
                auto e4m3 = [](int v) -> uint8_t {
                    if (v == 0) return 0;
                    const int sg = v < 0 ? 0x80 : 0, av = std::abs(v);
                    int e = 0;
                    while ((2 << e) <= av) e++;
                    return (uint8_t)(sg | ((e + 7) << 3) | ((av * 8) / (1 << e) - 8));
                };
                const uint64_t n_steps = 4ull * n_kgroups;
                s->narrow_pack1 = P == 1 && !(exp_int("KGWAS_NARROW_PACK", 1) == 0);  // (experiments: 0)
                std::vector<uint8_t> Bn(n_steps * 64 * 32, 0);
                std::vector<NarrowCol> ncols(P);
                std::vector<int> q(S);
                auto put_slot = [&](uint64_t slot, const std::vector<int>& v) {
                    for (uint64_t g = 0; g < n_kgroups; g++)
                        for (uint64_t jj = 0; jj < 4; jj++)
                            for (uint64_t k = 0; k < 128; k++) {
                                // FP4 side: k = 32 kb + 8 q + e' <-> bit 4 e' + jj of dword q of the lane's 16 bytes
                                const uint64_t kbA = k / 32, e = k % 32, smp = 512 * g + 128 * kbA + 32 * (e / 8) + 4 * (e % 8) + jj;
                                if (smp >= S) continue;
                                // FP8 side: lane kb = (k % 64) / 16, byte (k / 64) * 16 + k % 16
                                const uint64_t lane = slot + 16 * ((k % 64) / 16), byte = (k / 64) * 16 + k % 16;
                                Bn[((g * 4 + jj) * 64 + lane) * 32 + byte] = e4m3(v[smp]);
                            }
                };
                for (uint64_t j = 0; j < P; j++) {
                    const double Nd = (double)S, sum = (double)sums[j];
                    const double c = sum / Nd;
                    double mx = 0, A = 0;
                    std::vector<double> t(S);
                    for (uint64_t i = 0; i < S; i++) {
                        t[i] = (double)s->Y[j * S + i] - c;
                        mx = std::max(mx, std::fabs(t[i]));
                        A += std::fabs((double)s->Y[j * S + i]);
                    }
                    double u = mx > 0 ? mx / 15.0 : 1.0;
                    NarrowCol& nc = ncols[j];
                    for (int k = 0; k < NARROW_SLICES; k++) {
                        for (uint64_t i = 0; i < S; i++) {
                            int v = (int)std::lrint(t[i] / u);
                            v = std::max(-15, std::min(15, v));
                            q[i] = v;
                            t[i] -= u * v;
                        }
                        put_slot(4 * j + k, q);  // operand row 4 p + k; 4 p + 3 = ones
                        if (s->narrow_pack1)    // (one column: the same rows in the other three column slots, kernels.h NarrowArgs::pack1)
                            for (uint64_t t = 1; t < 4; t++) put_slot(4 * t + k, q);
                        nc.w[k] = 2.0 * u;
                        u /= 30.0;
                    }
                    double rpos = 0, rneg = 0, rmax = 0;
                    for (uint64_t i = 0; i < S; i++) {
                        if (s->dbg_keep_resid) s->dbg_resid[2][j * S + i] = t[i];
                        if (t[i] > 0) rpos += t[i]; else rneg -= t[i];
                        rmax = std::max(rmax, std::fabs(t[i]));
                    }
                    // (the slice products u_k * v and the running residual are evaluated in double: pad by their rounding)
                    const double fuzz = 64.0 * std::ldexp(1.0, -52) * (mx + std::fabs(c));
                    nc.t1 = Nd * c - sum;
                    nc.eg = (gamma * A * (1.0 + 1e-6) + 1e-12 * (1.0 + A)) * (1.0 + 1e-9);
                    nc.rall = (std::max(rpos, rneg) + Nd * fuzz) * (1.0 + 1e-9);
                    nc.rmax = (rmax + fuzz) * (1.0 + 1e-9);
                    // both sides evaluate N * x - N1 * sum and the slice sums in double: absolute slack of a few ulps of
                    // the largest intermediate (N * N * max|y|)
                    nc.pad = 256.0 * std::ldexp(1.0, -52) * Nd * Nd * (mx + std::fabs(c) + 1.0) + 1e-300;
                    // float32 pre-screen (score_narrow.hip): |r| <= N |ycf| (1 + 2^-10) + slackf: N1 |t1|, N E and the pad
                    // of the exact test, and the float32 roundings of the three products and sums.
                    {
                        const double slack = Nd * (nc.eg + std::min(nc.rall, Nd * nc.rmax)) + Nd * std::fabs(nc.t1) + nc.pad +
                                             Nd * std::ldexp(1.0, -20) * mx * Nd;
                        for (int k = 0; k < 3; k++) nc.wf[k] = (float)nc.w[k];
                        nc.slackf = std::nextafter((float)(slack * 1.001), std::numeric_limits<float>::infinity());
                    }
                }
                for (uint64_t j = 0; j < (s->narrow_pack1 ? 4 : P); j++) put_slot(4 * j + 3, std::vector<int>(S, 1));
                s->d_Bn.alloc(Bn.size());
                s->d_ncols.alloc(P);
                KGWAS_HIP(hipMemcpy(s->d_Bn.p, Bn.data(), Bn.size(), hipMemcpyHostToDevice));
                KGWAS_HIP(hipMemcpy(s->d_ncols.p, ncols.data(), P * sizeof(NarrowCol), hipMemcpyHostToDevice));

            }
            // ---- block-scaled filter (score_mx.hip), the default: FP6 (+ FP4 / FP6) slices on the integer grids
            //   A6 = {0..15, 16..30 step 2, 32..60 step 4} (E2M3 x 8),  A4 = {0, 1, 2, 3, 4, 6, 8, 12} (E2M1 x 2):
            //   y_i - c ~ w * t_i,  t_i = a6_i (one slice), 8 a6_i + a4_i (FP4 second slice) or 32 a6_i + a6'_i (FP6 second
            //   slice); the accumulator is kappa * sum_i g_i t_i, kappa = 1/16, 1/4, 1/16, so one accumulator unit is
            //   u = w / kappa phenotype units and everything above (kalpha, the error terms in units of Dc) carries over
            //   with that u. The ones column has t = 1 / kappa: its accumulator is N1.
            // Which filter (KGWAS_COARSE_MX=1|0 forces one): the block-scaled one unless its operands (1.25 bytes per sample
            // and column with two slices) make a row pass through more than ONE more LDS group than the int8 filter's single
            // slice (1 byte) does - every row is loaded, expanded and tested once per group. Measured, all kernels per 100 M
            // rows: 1024 x 101 (one group each) 14.2 ms against 15.2; 1135 x 101 (two each) 17.6 against 21.1; 2048 x 201
            // (five equal groups of three column tiles in one launch against four groups of four int8 tiles + the two-slice
            // ramp) 50.1 against 52.2 (51.5 with the int8 one-slice set + a block-scaled ramp, the arrangement beyond).
            bool use_mx;
            if (const char* e = opt_str("KGWAS_COARSE_MX")) {
                use_mx = atoi(e) != 0;
            } else {
                auto groups_for = [&](uint32_t tmax) {
                    uint64_t g = 1;
                    while (tmax && ((P + g - 1) / g + 1 + 15) / 16 > tmax) g++;
                    return tmax ? g : ~0ull;
                };
                const uint32_t steps = 4u * (uint32_t)(S / 512) + (uint32_t)((S % 512 + 127) / 128);
                uint32_t ctm = 0;
                for (uint32_t ct = 7; ct >= 1 && !ctm; ct--)
                    if (mx_lds_bytes(steps, ct, 2, 0) <= 160u * 1024u) ctm = ct;
                // (streamed operands - where that form is taken, see mxs_want above - are one group whatever the shape)
                const bool will_stream = mxs_can && (mxs_want >= 3 || !ctm || (groups_for(ctm) >= 2 && (mxs_want >= 2 || ctm <= 2)));
                // (beyond 5120 accessions there is no int8 plan to compare with: coarse_T = 0)
                use_mx = will_stream || !s->coarse_T || groups_for(ctm) <= groups_for(s->coarse_T) + 1;
            }
            // Where the int8 filter keeps the shape, its TWO-slice set (the ramp: the first chunks of a scan, many
            // candidates per row) is still the block-scaled one: at 2048 x 201 that is 13 column tiles x 2 slices x 16
            // K = 128 steps against 28 int8 tile-slices x 32 K = 64 steps - about half the matrix work per row for the same
            // ~1 survivor per candidate - and the chunks stay on it longer before the one-slice int8 set takes over
            // (pick_coarse_mode prices both sets in int8 tile-slice equivalents).
            const bool mixed = !use_mx && !s->narrow && !opt_set("KGWAS_COARSE_MX") && !opt_set("KGWAS_COARSE_SLICES") &&
                               want[0] && want[1] && !(exp_int("KGWAS_COARSE_MIXED", 1) == 0);
            if (use_mx && !s->narrow && !opt_set("KGWAS_COARSE_SLICES")) want[0] = false;  // one FP6 slice alone: only on request
            auto build_mx = [&](int mi) {
                const int ns = mi + 1;
                kgwas_scan::CoarseMode& M = s->cmode[mi];
                std::vector<int> G6, G4;  // the signed grids, ascending
                for (int q = 60; q >= 32; q -= 4) G6.push_back(-q);
                for (int q = 30; q >= 16; q -= 2) G6.push_back(-q);
                for (int q = 15; q >= -15; q--) G6.push_back(-q);
                for (int q = 16; q <= 30; q += 2) G6.push_back(q);
                for (int q = 32; q <= 60; q += 4) G6.push_back(q);
                for (int h : {-12, -8, -6, -4, -3, -2, -1, 0, 1, 2, 3, 4, 6, 8, 12}) G4.push_back(h);
                auto nearest = [](const std::vector<int>& g, double v) {  // index of the grid value closest to v
                    size_t hi = std::lower_bound(g.begin(), g.end(), v, [](int a, double b) { return (double)a < b; }) - g.begin();
                    if (hi == 0) return (size_t)0;
                    if (hi == g.size()) return g.size() - 1;
                    return (v - (double)g[hi - 1] <= (double)g[hi] - v) ? hi - 1 : hi;
                };
                auto e2m3 = [](int q) -> uint32_t {  // E2M3 code of q / 8
                    const uint32_t sg = q < 0 ? 0x20u : 0u;
                    const int a = std::abs(q);
                    if (a < 8) return sg | (uint32_t)a;
                    int e = 1, base = 8;
                    while (a >= 2 * base) base *= 2, e++;
                    return sg | ((uint32_t)e << 3) | (uint32_t)((a - base) / (base / 8));
                };
                auto e2m1 = [](int h) -> uint32_t {  // E2M1 code of h / 2
                    static const int tab[8] = {0, 1, 2, 3, 4, 6, 8, 12};
                    uint32_t i = 0;
                    while (tab[i] != std::abs(h)) i++;
                    return (h < 0 ? 8u : 0u) | i;
                };
                // whole 512-sample groups (the kernel reads all 64 bytes of those without a bounds check) + up to four quarter groups
                const uint32_t n_full = (uint32_t)(S / 512), nq = (uint32_t)((S % 512 + 127) / 128);
                const uint32_t n_steps = 4 * n_full + nq;
                // column tiles per LDS group the LDS can hold, and the LDS groups that takes, for a second-slice format
                auto ct_max = [&](uint32_t fp6) {
                    for (uint32_t ct = 7; ct >= 1; ct--)
                        if (mx_lds_bytes(n_steps, ct, (uint32_t)ns, fp6) <= 160u * 1024u) return ct;
                    return 0u;
                };
                auto groups_for = [&](uint32_t ctm) {
                    uint64_t g = 1;
                    while (((P + g - 1) / g + 1 + 15) / 16 > ctm) g++;
                    return g;
                };
                // Second slice: FP4. An FP6 one (1.1 survivors per candidate instead of 1.4) costs LDS, a slower MFMA (8.25
                // against 9.5 POP/s) and two more operand registers per tile: measured at 1135 x 101, the same 2 x 4 tiles,
                // filter 14.4 against 13.6 ms per 100 M rows and all kernels 18.3 against 17.8. KGWAS_MX_S1=6 selects it
                // (tests keep that kernel form covered).
                uint32_t s1_fp6 = 0;
                if (const char* e = opt_str("KGWAS_MX_S1")) s1_fp6 = ns == 2 && atoi(e) == 6 && ct_max(1) ? 1u : 0u;
                const uint32_t CTmax = ct_max(s1_fp6);
                // Operand-streaming form (score_mxs.hip): every row is loaded and expanded ONCE per operand group of up to 14 column
                // tiles, whatever the number of accessions; taken where the resident plan would pass every row through several LDS
                // groups. Up to 7 tiles (111 columns + the ones column): one column group, eight waves of 64 rows each. Beyond: TWO
                // column groups of up to 7 tiles per block (222 columns), the waves w and w + 4 working on the same 64 rows - each
                // group with its own ones column - and as many such operand groups (grid blocks sharing rows) as the columns need.
                // KGWAS_MXS_FORM=1 / 2: one column group of up to 13 tiles, eight waves of 32 rows / four waves of 64 rows.
                uint64_t stream_groups = 0, stream_ct = 0, stream_ng = 1;
                uint32_t stream_form = 0;
                if (mxs_can && ns == 2 && !s1_fp6 && !s->narrow) {
                    const int form_env = (int)opt_int("KGWAS_MXS_FORM", 0);
                    uint64_t g = 1, ng = 1, ct = 0;
                    if (P + 1 <= 7 * 16) {
                        ct = std::max<uint64_t>(3, (P + 1 + 15) / 16);
                    } else if (form_env == 1 || form_env == 2) {
                        while (((P + g - 1) / g + 1 + 15) / 16 > 13) g++;
                        ct = ((P + g - 1) / g + 1 + 15) / 16;
                        if (ct > 7) stream_form = (uint32_t)form_env;
                    } else {
                        ng = 2;
                        g = 2;
                        while (((P + g - 1) / g + 1 + 15) / 16 > 7) g += 2;
                        ct = std::max<uint64_t>(4, ((P + g - 1) / g + 1 + 15) / 16);
                    }
                    const bool take = mxs_want >= 3 || !CTmax || (groups_for(CTmax) >= 2 && (mxs_want >= 2 || CTmax <= 2));
                    if (take && mxs_supported((uint32_t)ct, (uint32_t)ng, 2, 0) && mxs_lds_bytes((uint32_t)ct, (uint32_t)ng) <= 160u * 1024u)
                        stream_groups = g, stream_ct = ct, stream_ng = ng;
                }
                if (!CTmax && !stream_groups) throw Error(KGWAS_ERR_ARG, "coarse filter: too many accessions for the LDS");
                const int sh = ns == 1 ? 0 : (s1_fp6 ? 5 : 3);                 // t = 2^sh * a6 + a1
                const double kappa = (ns == 2 && !s1_fp6) ? 0.25 : 0.0625;     // accumulator = kappa * sum g t
                const int t_ones = (int)(1.0 / kappa);                         // in the LAST slice (a6 = 0 with two slices)
                const double t_max = ns == 1 ? 60.0 : (s1_fp6 ? 32.0 * 60.0 + 60.0 : 8.0 * 60.0 + 12.0);
                const std::vector<int>& G1 = s1_fp6 ? G6 : G4;
                std::vector<int> a0(S), a1(S);  // (the serial callers' scratch)
                auto quantise_mx = [&](uint64_t j, CoarseCol& cc, ErrBound& eb, std::vector<int>& a0, std::vector<int>& a1) {
                    const double Nd = (double)S, sum = (double)sums[j];
                    const double c = sum / Nd;
                    double mx = 0, A = 0;
                    for (uint64_t i = 0; i < S; i++) {
                        const double y = (double)s->Y[j * S + i];
                        mx = std::max(mx, std::fabs(y - c));
                        A += std::fabs(y);
                    }
                    const double w = mx > 0 ? mx / t_max : 1.0;
                    const double u = w / kappa;  // one accumulator unit in phenotype units
                    double rpos = 0, rneg = 0, rmax = 0;
                    for (uint64_t i = 0; i < S; i++) {
                        const double y = (double)s->Y[j * S + i] - c;
                        const double x = y / w;
                        int b0 = 0, b1 = 0;
                        if (ns == 1) {
                            b0 = G6[nearest(G6, x)];
                        } else {
                            // the first slice's neighbours of x / 2^sh, each with its best second slice
                            const double sc = (double)(1 << sh);
                            const size_t k0 = nearest(G6, x / sc);
                            double best = 1e300;
                            for (size_t k = k0 ? k0 - 1 : 0; k <= std::min(k0 + 1, G6.size() - 1); k++) {
                                const int c1 = G1[nearest(G1, x - sc * G6[k])];
                                const double r = std::fabs(x - sc * G6[k] - c1);
                                if (r < best) best = r, b0 = G6[k], b1 = c1;
                            }
                        }
                        a0[i] = b0;
                        a1[i] = b1;
                        const double r = y - w * ((double)(1 << sh) * b0 + b1);
                        if (s->dbg_keep_resid) s->dbg_resid[ns - 1][j * S + i] = r;
                        if (r > 0) rpos += r; else rneg -= r;
                        rmax = std::max(rmax, std::fabs(r));
                    }
                    const double rho = Nd * std::fabs(Nd * c - sum) * 2.0 + 1e-9 * (1.0 + std::fabs(sum));
                    const double Eg = gamma * A * (1.0 + 1e-6) + 1e-12 * (1.0 + A);
                    cc.kalpha = (1.0 - std::ldexp(1.0, -19)) / (Nd * u);
                    cc.iu = up(1.0 / u * (1.0 + 1e-6));
                    eb.eg = up((Eg + rho / Nd) * (1.0 + 1e-6) + 1e-30);
                    eb.rall = up(std::max(rpos, rneg) * (1.0 + 1e-6));
                    eb.rmax = up(rmax * (1.0 + 1e-6));
                    const double iu = 1.0 / u * (1.0 + 1e-6);
                    eb.egD = up((double)eb.eg * iu);
                    eb.rallD = up((double)eb.rall * iu);
                    eb.rmaxD = up((double)eb.rmax * iu);
                };
                // LDS groups: as few as hold all columns (+ a ones column each); or groups filled to the last slot and one
                // smaller launch for the rest when that multiplies fewer tiles
                uint64_t n_lgroups = stream_groups ? stream_groups : groups_for(CTmax);
                uint64_t cper = (P + n_lgroups - 1) / n_lgroups;
                struct Plan {
                    uint64_t j0, n, CT, groups, cper;
                };
                std::vector<Plan> plan;
                plan.push_back(Plan{0, P, stream_groups ? stream_ct : (cper + 1 + 15) / 16, n_lgroups, cper});
                if (n_lgroups > 1 && !stream_groups) {
                    const uint64_t cpf = (uint64_t)CTmax * 16 - 1;
                    const uint64_t full = P / cpf, rem = P - full * cpf;
                    const uint64_t CTr = rem ? (rem + 1 + 15) / 16 : 0;
                    // (a second launch for the rest is worth two and a half tiles of its own: its few column tiles multiply at
                    // a fraction of the full groups' efficiency. 1135 x 101: 2 x 4 tiles in one launch 13.6 ms per 100 M rows,
                    // 6 + 1 tiles in two 16.3; 2048 x 201: 5 x 3 tiles in one launch 39.2, 4 x 3 + 1 in two 40.2)
                    const bool no_split = exp_set("KGWAS_COARSE_NOSPLIT");  // experiments
                    if (full >= 1 && full + (rem ? 1 : 0) <= n_lgroups && 2 * (full * CTmax + CTr) + 5 < 2 * n_lgroups * plan[0].CT && !no_split) {
                        plan.clear();
                        plan.push_back(Plan{0, full * cpf, CTmax, full, cpf});
                        if (rem) plan.push_back(Plan{full * cpf, rem, CTr, 1, rem});
                    }
                }
                M.mx = true;
                M.mx_full = n_full;
                M.mx_quarter = nq;
                M.mx_s1_fp6 = s1_fp6;
                M.mx_scale0 = 0x01010101u * (uint32_t)(0x7F + (ns == 1 ? 0 : 5));
                M.slices = (uint32_t)ns;
                M.n_parts = (uint32_t)plan.size();
                M.tile_slices = 0;
                uint32_t groups_all = 0;
                const uint32_t SB = mx_step_bytes_rt((uint32_t)ns, s1_fp6);
                for (size_t pi = 0; pi < plan.size(); pi++) {
                    const Plan& pl = plan[pi];
                    kgwas_scan::CoarsePart& Pt = M.part[pi];
                    const uint32_t CT = (uint32_t)pl.CT, slots = CT * 16;
                    Pt.T = CT;
                    Pt.n_lgroups = (uint32_t)pl.groups;
                    const uint64_t NGs = stream_groups ? stream_ng : 1;  // column groups per block (streaming form)
                    if (stream_groups) {
                        Pt.n_lgroups = (uint32_t)(pl.groups / NGs);  // operand groups = grid blocks per row block
                        Pt.ng = (uint32_t)NGs;
                        Pt.stream = 1u + stream_form;
                        s->st.coarse_mx_stream = Pt.stream;
                    }
                    M.tile_slices += CT * (uint32_t)ns * (uint32_t)pl.groups;
                    groups_all += Pt.n_lgroups;
                    const size_t group_bytes = (size_t)n_steps * CT * SB;
                    std::vector<uint8_t> Bq(pl.groups * group_bytes + 1024, 0);  // (the streaming form's last transfer of a slab reads up to 1 KB past it)
                    std::vector<CoarseCol> cols(pl.groups * slots);
                    for (auto& cc : cols) {
                        memset(&cc, 0, sizeof(cc));
                        cc.pheno = -1;
                    }
                    // the slice values of operand column `slot` of LDS group lg: v0 on the A6 grid, v1 on the second slice's
                    auto put = [&](uint64_t lg, uint64_t slot, const std::vector<int>& v0, const std::vector<int>& v1) {
                        const uint64_t t = slot / 16, n = slot % 16;
                        for (uint64_t st = 0; st < n_steps; st++) {
                            // (streaming form: the NG column groups of a block lie side by side within a step's slab)
                            uint8_t* blk = &Bq[(lg / NGs) * (NGs * group_bytes) + ((st * NGs + lg % NGs) * CT + t) * SB];
                            for (uint64_t kb = 0; kb < 4; kb++) {
                                const uint64_t lane = kb * 16 + n;
                                for (uint64_t e = 0; e < 32; e++) {
                                    // score_mx.hip: k = 32 kb + e <-> sample
                                    const uint64_t smp = st < 4ull * n_full ? 512 * (st / 4) + 128 * kb + 32 * (e / 8) + 4 * (e % 8) + st % 4
                                                                            : 512ull * n_full + 128 * (st - 4ull * n_full) + 32 * kb + 4 * (e % 8) + e / 8;
                                    if (smp >= S) continue;
                                    auto put6 = [&](uint8_t* part, int q) {  // 6-bit field e of the lane's 6 dwords: dwords 0-3 | 4-5
                                        const uint32_t code = e2m3(q);
                                        for (int b = 0; b < 6; b++)
                                            if (code & (1u << b)) {
                                                const uint64_t bit = 6 * e + b, dw = bit / 32;
                                                uint8_t* d = dw < 4 ? part + lane * 16 + dw * 4 : part + 1024 + lane * 8 + (dw - 4) * 4;
                                                d[(bit % 32) / 8] |= (uint8_t)(1u << (bit % 8));
                                            }
                                    };
                                    put6(blk, v0[smp]);
                                    if (ns == 2) {
                                        if (s1_fp6)
                                            put6(blk + 1536, v1[smp]);
                                        else
                                            blk[1536 + lane * 16 + e / 2] |= (uint8_t)(e2m1(v1[smp]) << (4 * (e % 2)));
                                    }
                                }
                            }
                        }
                    };
                    // The columns on a few threads (quantising and packing 101 columns of 1135 samples took 35 ms of a session's
                    // creation - a tenth of a whole `associate_kmers` run on a 6 GB table): a column's operand bytes are its own
                    // (lane kb * 16 + slot % 16 of tile slot / 16), as are its constants; the bounds are folded afterwards.
                    {
                        std::vector<ErrBound> ebs(pl.n);
                        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(usable_cpus(), 16), pl.n / 4));
                        std::atomic<uint64_t> next(0);
                        kgwas_run_on_threads(nt, "kgwas-quant", [&] {
                            std::vector<int> b0(S), b1(S);
                            try {
                                for (uint64_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < pl.n;) {
                                    const uint64_t j = pl.j0 + i, lg = i / pl.cper, slot = i % pl.cper;
                                    CoarseCol& cc = cols[lg * slots + slot];
                                    quantise_mx(j, cc, ebs[i], b0, b1);
                                    cc.pheno = (int32_t)j;
                                    put(lg, slot, b0, b1);
                                }
                            } catch (...) {
                                next.store(pl.n);  // (the other threads stop at their next column)
                                throw;
                            }
                        });
                        for (const ErrBound& eb : ebs) {
                            M.eg_max = std::max(M.eg_max, eb.egD);
                            M.rall_max = std::max(M.rall_max, eb.rallD);
                            M.rmax_max = std::max(M.rmax_max, eb.rmaxD);
                        }
                    }
                    {  // ones column: accumulator = N1
                        std::vector<int> ones(S, t_ones), zeros(S, 0);
                        for (uint64_t lg = 0; lg < pl.groups; lg++) put(lg, slots - 1, ns == 1 ? ones : zeros, ones);
                    }
                    Pt.d_Bq.alloc(Bq.size());
                    Pt.d_cols.alloc(cols.size());
                    KGWAS_HIP(hipMemcpy(Pt.d_Bq.p, Bq.data(), Bq.size(), hipMemcpyHostToDevice));
                    KGWAS_HIP(hipMemcpy(Pt.d_cols.p, cols.data(), cols.size() * sizeof(CoarseCol), hipMemcpyHostToDevice));
                }
                s->st.coarse_mode_tiles[mi] = M.part[0].T;
                s->st.coarse_mode_lgroups[mi] = groups_all;
                s->st.coarse_mode_tile_slices[mi] = M.tile_slices;
                if (use_mx) s->st.coarse_mx = 1;
                s->st.coarse_mx_s1_fp6 = s1_fp6;
                s->st.coarse_mx_steps = n_steps;
                // the set's matrix work per row in int8 tile-slice equivalents (a K = 128 step is one of the 8 n_kgroups K = 64
                // steps' worth of two; measured 30 % less efficient per MFMA with three column tiles per LDS group: 0.48 against 0.37 ms per M rows at 2048 x 201)
                M.tile_slices_eq = (double)M.tile_slices * (double)n_steps / (8.0 * (double)n_kgroups) * ((CTmax <= 3 && !stream_groups) ? 1.30 : 1.0);
                M.ready = true;
            };
            for (int mi = 0; mi < 2; mi++) {
                if (!want[mi]) continue;
                if (use_mx || (mixed && mi == 1)) {
                    build_mx(mi);
                    continue;
                }
                const int ns = mi + 1;
                kgwas_scan::CoarseMode& M = s->cmode[mi];
                // Operand columns ("slots") per LDS group: the group's share of the phenotype columns, padding, and
                // the ones column in the last slot (its dot product is the row's masked popcount N1).
                uint32_t Tmax = s->coarse_T;  // largest tile count whose operands fit the LDS
                if (ns == 2) Tmax &= ~1u;
                uint64_t n_lgroups = 1, cper = P;
                uint32_t T = 0;
                for (;; n_lgroups++) {
                    cper = (P + n_lgroups - 1) / n_lgroups;  // phenotype columns per group
                    T = (uint32_t)(ns * ((cper + 1 + 15) / 16));
                    if (T <= Tmax) break;
                }
                // plan[i] = {first column, columns, T, LDS groups, columns per group}
                struct Plan {
                    uint64_t j0, n, T, groups, cper;
                };
                std::vector<Plan> plan;
                plan.push_back(Plan{0, P, T, n_lgroups, cper});
                if (n_lgroups > 1) {
                    // The balanced split pads every group (201 columns, 4 tiles per group: 4 x (51 + ones) of 4 x 64
                    // slots = 16 tiles for 13 tiles' worth of columns). Alternative: groups filled to the last slot and
                    // ONE smaller launch for the rest - taken when it multiplies fewer tiles with no more row passes.
                    const uint64_t cpf = (uint64_t)(Tmax / (uint32_t)ns) * 16 - 1;  // columns of a full group
                    const uint64_t full = P / cpf, rem = P - full * cpf;
                    const uint64_t Tr = rem ? (uint64_t)ns * ((rem + 1 + 15) / 16) : 0;
                    static const bool no_split = exp_set("KGWAS_COARSE_NOSPLIT");  // experiments
                    if (full >= 1 && full + (rem ? 1 : 0) <= n_lgroups && full * Tmax + Tr < n_lgroups * T && !no_split) {
                        plan.clear();
                        plan.push_back(Plan{0, full * cpf, Tmax, full, cpf});
                        if (rem) plan.push_back(Plan{full * cpf, rem, Tr, 1, rem});
                    }
                }
                M.slices = (uint32_t)ns;
                M.n_parts = (uint32_t)plan.size();
                M.tile_slices = 0;
                uint32_t groups_all = 0;
                for (size_t pi = 0; pi < plan.size(); pi++) {
                    const Plan& pl = plan[pi];
                    kgwas_scan::CoarsePart& Pt = M.part[pi];
                    const uint32_t Tp = (uint32_t)pl.T;
                    const uint32_t PG = Tp / (uint32_t)ns, slots = PG * 16;
                    Pt.T = Tp;
                    Pt.n_lgroups = (uint32_t)pl.groups;
                    M.tile_slices += Tp * (uint32_t)pl.groups;
                    groups_all += (uint32_t)pl.groups;
                    std::vector<int8_t> Bq(pl.groups * n_kgroups * 8ull * Tp * 1024ull, 0);
                    std::vector<CoarseCol> cols(pl.groups * slots);
                    for (auto& cc : cols) {
                        memset(&cc, 0, sizeof(cc));
                        cc.pheno = -1;
                    }
                    auto put = [&](uint64_t lg, uint64_t slot, const std::vector<int>& v0, const std::vector<int>& v1) {
                        const uint64_t pgl = slot / 16, n = slot % 16;
                        for (uint64_t g = 0; g < n_kgroups; g++)
                            for (uint64_t jj = 0; jj < 8; jj++)
                                for (uint64_t kg = 0; kg < 4; kg++)
                                    for (uint64_t e = 0; e < 16; e++) {
                                        // k-element e of step jj <-> sample (score_coarse.hip: expand_step)
                                        const uint64_t smp = 512 * g + 128 * kg + 32 * (e / 4) + 8 * (e % 4) + jj;
                                        if (smp >= S) continue;
                                        const uint64_t lane = kg * 16 + n;
                                        const uint64_t base = (((lg * n_kgroups + g) * 8 + jj) * Tp);
                                        if (ns == 1) {
                                            Bq[((base + pgl) * 64 + lane) * 16 + e] = (int8_t)v0[smp];
                                        } else {
                                            Bq[((base + 2 * pgl) * 64 + lane) * 16 + e] = (int8_t)v0[smp];
                                            Bq[((base + 2 * pgl + 1) * 64 + lane) * 16 + e] = (int8_t)v1[smp];
                                        }
                                    }
                    };
                    for (uint64_t j = pl.j0; j < pl.j0 + pl.n; j++) {
                        const uint64_t lg = (j - pl.j0) / pl.cper, slot = (j - pl.j0) % pl.cper;
                        CoarseCol& cc = cols[lg * slots + slot];
                        ErrBound eb;
                        quantise(j, ns, cc, eb);
                        M.eg_max = std::max(M.eg_max, eb.egD);
                        M.rall_max = std::max(M.rall_max, eb.rallD);
                        M.rmax_max = std::max(M.rmax_max, eb.rmaxD);
                        cc.pheno = (int32_t)j;
                        put(lg, slot, q0, q1);
                    }
                    {  // ones column: Dc = N1 (one slice: q0 = 1; two slices: Dc = 254*D0 + D1 with q0 = 0, q1 = 1)
                        std::vector<int> ones(S, 1), zeros(S, 0);
                        for (uint64_t lg = 0; lg < pl.groups; lg++) put(lg, slots - 1, ns == 1 ? ones : zeros, ones);
                    }
                    Pt.d_Bq.alloc(Bq.size());
                    Pt.d_cols.alloc(cols.size());
                    KGWAS_HIP(hipMemcpy(Pt.d_Bq.p, Bq.data(), Bq.size(), hipMemcpyHostToDevice));
                    KGWAS_HIP(hipMemcpy(Pt.d_cols.p, cols.data(), cols.size() * sizeof(CoarseCol), hipMemcpyHostToDevice));
                }
                s->st.coarse_mode_tiles[mi] = M.part[0].T;
                s->st.coarse_mode_lgroups[mi] = groups_all;
                s->st.coarse_mode_tile_slices[mi] = M.tile_slices;
                M.ready = true;
            }
            tcreate("operand sets built and uploaded");
            s->key_slots = (uint32_t)std::min<uint64_t>((uint64_t)s->cap * P, 0x7FFFFFFFull);
            s->d_surv_sorted.alloc(s->key_slots);
            s->bitmap_words = (s->chunk_max + 63) / 64;
            s->d_bitmap.alloc(P * s->bitmap_words);
            s->d_bm_blocks.alloc(P * ((s->bitmap_words + 1023) / 1024 + 1) + 4);
            if (!s->narrow) s->d_bm_mask.alloc(P * ((s->bitmap_words + 1023) / 1024) * 4 + 4);
            s->bitmap_clean = false;
            s->d_surv_cnt.alloc(P);
            s->d_surv_off.alloc(P);
            s->d_key_count.alloc(2);  // the survivor count; the narrow re-score kernel's block counter
            s->d_tile_pref.alloc(P + 1);
            s->d_tile_cnt.alloc((size_t)s->key_slots / 256 + P + 2);
            s->d_tmp_score.alloc(s->key_slots);
            // The record copies run as blit kernels (rocprofv3 shows __amd_rocclr_copyBuffer, not SDMA transfers), and at
            // normal priority they are only dispatched in the gaps of the compute stream: behind a 0.75 ms filter launch
            // of a one-column scan, a chunk's 1 MB of records reached the host 1.3-2.8 ms after its counts. A high-priority
            // queue gets them onto the chip between the running launch's workgroups. KGWAS_COPY_PRIO=0: the old behaviour.
            {
                int least = 0, greatest = 0;
                KGWAS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
                const bool hi = !(exp_int("KGWAS_COPY_PRIO", 1) == 0);
                KGWAS_HIP(hipStreamCreateWithPriority(&s->copy_stream, hipStreamNonBlocking, hi ? greatest : least));
            }
            s->row_key_bits = 1;
            while (s->row_key_bits < 32 && (1ull << s->row_key_bits) < s->chunk_max) s->row_key_bits++;
        }
        if (!s->direct) s->d_sq.alloc(s->chunk_max * 2 * W_m);

        {
            if (s->coarse) {
                // device side: 20 B x key_slots of HBM per slot, up to 4 GiB in all; host side: the record ring
                const uint64_t slot_bytes = (uint64_t)s->key_slots * 20;
                s->n_slots = (int)std::min<uint64_t>(MAX_SLOTS, std::max<uint64_t>(4, (4ull << 30) / std::max<uint64_t>(slot_bytes, 1)));
                // Pinning memory is slow - 1 GiB takes 0.21-0.25 s, two thirds of a session's creation (KGWAS_TRACE), and every
                // other allocation of the process queues behind it, so a thread of its own does not hide it: the ring is sized
                // for what the scan plans to have in flight instead of the cap. A chunk is planned to fill 40 % of the key list
                // (next_sparse_chunk) and the GPU runs up to ~16 chunks ahead of the replay: 20 planned chunks' records
                // (8 x a slot's worst case), at least 64 MiB and two worst-case chunks, at most 1 GiB - 390 MB at 101 columns
                // (top-10001), 775 MB at 201. A ring that fills up only makes the GPU wait for the replay (fetch_records).
                s->ring_size = (size_t)std::max<uint64_t>(std::min<uint64_t>(1ull << 30, std::min<uint64_t>((uint64_t)s->n_slots * slot_bytes, std::max<uint64_t>(64ull << 20, 8 * slot_bytes))),
                                                          2 * slot_bytes + 4096);
                // (tests: a ring barely larger than one chunk's worst case, so that it wraps and fills up)
                if (opt_str("KGWAS_RING_BYTES"))
                    s->ring_size = (size_t)std::max<uint64_t>(strtoull(opt_str("KGWAS_RING_BYTES"), nullptr, 10), slot_bytes + 4096);
                tcreate("device buffers and slots allocated");
                s->ring.alloc(s->ring_size);
                s->ring_dev = s->ring.dev();
                tcreate("pinned record ring allocated");
            } else {
                const uint64_t slot_bytes = (uint64_t)s->cap * P * sizeof(Cand);
                s->n_slots = (int)std::min<uint64_t>(16, std::max<uint64_t>(4, (1ull << 30) / std::max<uint64_t>(slot_bytes, 1)));
            }
        }
        for (int si = 0; si < s->n_slots + (s->coarse ? 1 : 0); si++) {
            const bool is_redo = si == s->n_slots;
            Slot& sl = is_redo ? s->redo : s->slot[si];
            if (s->coarse && !is_redo) {
                sl.d_so_score.alloc(s->key_slots);
                sl.d_so_kmer.alloc(s->key_slots);
                sl.d_so_row.alloc(s->key_slots);
                sl.d_meta.alloc(2 * P + 4);
                sl.h_meta.alloc(2 * P + 4);
                sl.h_thr.alloc(P);
                memset(sl.h_thr.p, 0, P * sizeof(double));
                memset(sl.h_meta.p, 0, (2 * P + 4) * sizeof(uint32_t));
                sl.h_meta_dev = sl.h_meta.dev();
                sl.h_thr_dev = sl.h_thr.dev();
                KGWAS_HIP(hipEventCreateWithFlags(&sl.ev_counts, hipEventBlockingSync));
            } else {
                sl.cand.alloc((uint64_t)s->cap * P);
                sl.d_cand = sl.cand.dev();
            }
            sl.d_cnt.alloc(P);
            sl.h_cnt.alloc(P);
            sl.d_tested.alloc(TESTED_SHARDS);
            sl.h_tested.alloc(TESTED_SHARDS);
            sl.h_tested_dev = sl.h_tested.dev();
            KGWAS_HIP(hipEventCreate(&sl.ev_sq0));
            KGWAS_HIP(hipEventCreate(&sl.ev_k0));
            KGWAS_HIP(hipEventCreate(&sl.ev_k1));
            // (blocking wait: the control thread sleeps instead of spinning beside the replay workers)
            KGWAS_HIP(hipEventCreateWithFlags(&sl.ev_done, hipEventBlockingSync));
            KGWAS_HIP(hipEventCreate(&sl.ev_mid));
        }
        s->d_dense.alloc(P * s->dense_rows);
        s->h_dense.alloc(P * s->dense_rows);
        s->d_n1.alloc(s->dense_rows);
        s->h_n1.alloc(s->dense_rows);
        s->d_kmer.alloc(s->dense_rows);
        s->h_kmer.alloc(s->dense_rows);
        s->h_dense_dev = s->h_dense.dev();
        s->h_n1_dev = s->h_n1.dev();
        s->h_kmer_dev = s->h_kmer.dev();
        s->d_tested_dense.alloc(TESTED_SHARDS);

        make_heaps(s.get());
        {
            const char* fr = opt_str("KGWAS_FULL_REPLAY");
            s->lazy_enabled = s->coarse && !s->record_history && !(fr && atoi(fr) != 0);
            s->lazy_log_mode = s->lazy_enabled && s->history_ring != 0;
            lazy_reset(s.get());
        }
        s->hist.resize(P);
        s->keys.resize(P);
        s->col_ms.assign(P, 0.0);
        tcreate("buffers done");
        s->trace = opt_set("KGWAS_TRACE");
        unsigned nt = p->host_threads ? p->host_threads : usable_cpus();
        if (const char* e = opt_str("KGWAS_HOST_THREADS"))
            if (atoi(e) > 0) nt = (unsigned)atoi(e);
        nt = (unsigned)std::min<uint64_t>(nt, P);
        s->pool.reset(new Pool(nt, pick_replay_cpus(nt, s->device)));
        s->st.replay_threads = nt;
        s->ingest.producer_cpus_ = p->host_threads ? p->host_threads : usable_cpus();
        // Column groups of the replay. Worker w owns the columns w, w + T, ... of the first floor(P / T) * T columns,
        // in groups of at most MAX_LOCKSTEP (a group's heaps take their replacements in lockstep, and stay in their
        // worker's cache from chunk to chunk); the P mod T columns left over float: each is a group of its own that
        // whichever worker is furthest ahead takes, which evens out what a static map cannot (101 columns on 16
        // workers is 5 x 7 + 11 x 6: the 7-column workers set the pace, 17 % above the mean).
        {
            const uint64_t T = nt, base = P / T;
            const uint64_t MKc = (uint64_t)BestHeap::MAX_LOCKSTEP;
            uint64_t per = base ? (base + ((base + MKc - 1) / MKc) - 1) / ((base + MKc - 1) / MKc) : 0;  // balanced split
            if (const char* e = exp_str("KGWAS_REPLAY_GROUP"))
                if (atoi(e) > 0 && per) per = std::min<uint64_t>((uint64_t)atoi(e), MKc);
            for (uint64_t w = 0; w < T && base; w++) {
                std::vector<uint32_t> cur;
                for (uint64_t i = 0; i < base; i++) {
                    cur.push_back((uint32_t)(i * T + w));
                    if (cur.size() == per || i + 1 == base) {
                        s->grp_cols.push_back(cur);
                        s->grp_home.push_back((int)w);
                        cur.clear();
                    }
                }
            }
            for (uint64_t j = base * T; j < P; j++) {
                s->grp_cols.push_back(std::vector<uint32_t>(1, (uint32_t)j));
                s->grp_home.push_back(-1);
            }
            s->n_groups0 = s->grp_cols.size();
            s->n_groups.store(s->n_groups0);
            s->grp_cols0 = s->grp_cols;
            const size_t cap = s->n_groups0 + (size_t)P;  // room for every column as a group of its own (split_group)
            s->grp_cols.resize(cap);
            s->gstate.reset(new kgwas_scan::GroupState[cap]);
            s->grp_owner.reset(new std::atomic<int>[cap]);
            s->col_popped.reset(new std::atomic<uint8_t>[(size_t)P]);
            for (uint64_t j = 0; j < P; j++) s->col_popped[j].store(0);
            s->res_kmer.resize(P);
            s->res_row.resize(P);
            s->res_score.resize(P);
            for (size_t g = 0; g < cap; g++) s->grp_owner[g].store(g < s->n_groups0 ? s->grp_home[g] : -1);
            if (const char* e = opt_str("KGWAS_SPLIT_LAGGING")) s->split_lagging = atoi(e) != 0;
            if (const char* e = opt_str("KGWAS_FLOAT_LEAD")) s->float_lead = (uint64_t)std::max(0, atoi(e));
            if (const char* e = opt_str("KGWAS_DEBUG_SLOW_WORKER")) {
                int w = -1, pct = 100, min_us = 0;
                if (sscanf(e, "%d:%d:%d", &w, &pct, &min_us) >= 1) {
                    s->dbg_slow_worker = w;
                    s->dbg_slow_pct = pct;
                    s->dbg_slow_min_us = min_us;
                }
            }
            s->slot_left.reset(new std::atomic<uint32_t>[MAX_SLOTS]);
            for (int i = 0; i < MAX_SLOTS; i++) s->slot_left[i].store(0);
            kgwas_scan* raw = s.get();
            s->rp_fn = [raw](size_t w) { replay_worker(raw, w); };
        }
        s->st.kernel_used = s->narrow ? (uint32_t)KGWAS_KERNEL_NARROW : s->coarse ? (uint32_t)KGWAS_KERNEL_COARSE : kern;
        s->st.direct_mode = s->direct ? 1 : 0;
        tcreate("pool, heaps and arena ready");
        *out = s.release();
    });
}

}  // extern "C"
