// tools/probe_score.hip — developer probe (not part of the product): times score_mfma_kernel on
// random rows with compile-time ablations to locate the bottleneck (ablate-before-optimizing).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I kmersgwas_amd/csrc [-DKGWAS_ABLATE=n] tools/probe_score.hip -o probe
#include "../kmersgwas_amd/csrc/score_mfma.hip"
#include "../kmersgwas_amd/csrc/aux_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace kgwas;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char** argv) {
    const uint64_t n_rows = argc > 1 ? strtoull(argv[1], 0, 10) : (4ull << 20);
    const uint32_t S = argc > 2 ? atoi(argv[2]) : 1024, P = argc > 3 ? atoi(argv[3]) : 101;
    const uint32_t rpb = argc > 4 ? atoi(argv[4]) : 1024;
    const uint32_t W_f = (S + 63) / 64, W_m = 2 * ((S + 127) / 128), L = 64 * W_m, nct = (P + 15) / 16;
    uint64_t* d_rows;
    CK(hipMalloc(&d_rows, n_rows * (1 + W_f) * 8));
    CK(launch_synth(d_rows, 0, n_rows, S, 1234, 0));
    std::vector<float> Y((size_t)nct * L * 16);
    for (auto& v : Y) v = (float)rand() / RAND_MAX - 0.5f;
    std::vector<uint32_t> dmask(2 * W_m, 0xFFFFFFFFu);
    std::vector<float> sums(P, 1.0f);
    std::vector<double> thr(P, 1e30);
    float *d_Y, *d_sums; uint32_t* d_dmask; double* d_thr; Cand* d_cand; uint32_t* d_cnt; unsigned long long* d_tested;
    CK(hipMalloc(&d_Y, Y.size() * 4)); CK(hipMemcpy(d_Y, Y.data(), Y.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_sums, P * 4)); CK(hipMemcpy(d_sums, sums.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_dmask, dmask.size() * 4)); CK(hipMemcpy(d_dmask, dmask.data(), dmask.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_thr, P * 8)); CK(hipMemcpy(d_thr, thr.data(), P * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cand, (size_t)P * 1024 * sizeof(Cand))); CK(hipMalloc(&d_cnt, P * 4)); CK(hipMalloc(&d_tested, 8));
    CK(hipMemset(d_cnt, 0, P * 4)); CK(hipMemset(d_tested, 0, 8));
    ScoreArgs a{};
    a.src.base = (const uint32_t*)d_rows; a.src.stride_dw = 2 * (1 + W_f); a.src.off_dw = 2; a.src.avail_dw = 2 * W_f;
    a.dmask = d_dmask; a.file_rows = d_rows; a.file_stride_w = 1 + W_f; a.n_rows = n_rows; a.first_row = 0;
    a.S = S; a.W_m = W_m; a.n_pheno = P; a.min_count = (uint32_t)(S * 0.05 + 0.999);
    a.Ymfma = d_Y; a.sums = d_sums; a.thr = d_thr; a.cand = d_cand; a.cand_cnt = d_cnt; a.cap = 1024; a.tested = d_tested;
    uint32_t nb_full = 0; while (nb_full < W_m / 2 && 4 * nb_full + 3 < 2 * W_f) nb_full++;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; it++) CK(launch_score_mfma(a, rpb, nb_full, 0));
    CK(hipDeviceSynchronize());
    const int iters = 5;
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; it++) CK(launch_score_mfma(a, rpb, nb_full, 0));
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double flop_pad = 2.0 * n_rows * S * nct * 16, flop = 2.0 * n_rows * S * P;
#ifndef KGWAS_ABLATE
#define KGWAS_ABLATE 0
#endif
    printf("ablate=%d rows=%llu S=%u P=%u rpb=%u : %.3f ms  useful %.1f TF/s  padded %.1f TF/s (%.1f%% of 157.3)\n", KGWAS_ABLATE,
           (unsigned long long)n_rows, S, P, rpb, ms, flop / ms / 1e9, flop_pad / ms / 1e9, flop_pad / ms / 1e9 / 157.3 * 100);
    return 0;
}
