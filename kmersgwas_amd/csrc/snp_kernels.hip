// snp_kernels.hip — device side of the SNP twin of the scorer (src/snps_multiple_databases.cpp).
//
// A PLINK .bed holds two bits per sample and SNP. The reference turns every SNP into three bit planes over the
// phenotyped samples (:112-146): "presence" (A/A), "non-missing" and "heterozygous", and scores it with three
// dot_product_SSE4 chains (:38-63) - the same four-lane float32 order as calculate_kmer_score - plus a double tail
// (calculate_grammmar_approx_association, :155-172). Both steps run here: snp_planes_kernel builds the planes from
// the raw .bed bytes, snp_score_kernel reproduces the chains with select-and-add (one lane per SNP, phenotype
// values wave-uniform) and the tail without FMA contraction, so scores are bit-identical to the reference's.
#include "kernels.h"

namespace kgwas {

namespace {

// planes[snp][3][ndw]: 0 = presence (dubit 3), 1 = non-missing (dubit 0, 2, 3), 2 = heterozygous (dubit 2).
__global__ void __launch_bounds__(256) snp_planes_kernel(const uint8_t* bed, uint64_t n_snps, uint32_t bytes_per_snp,
                                                         const uint32_t* byte_idx, const uint32_t* shift, uint32_t S, uint32_t ndw,
                                                         uint32_t* planes) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_snps * ndw) return;
    const uint64_t snp = i / ndw;
    const uint32_t d = (uint32_t)(i - snp * ndw);
    const uint8_t* row = bed + snp * bytes_per_snp;
    uint32_t pres = 0, tot = 0, het = 0;
    for (uint32_t b = 0; b < 32; b++) {
        const uint32_t si = 32u * d + b;
        if (si >= S) break;
        const uint32_t dubit = (row[byte_idx[si]] >> shift[si]) & 3u;  // :130-131
        pres |= (dubit == 3u ? 1u : 0u) << b;
        tot |= (dubit != 1u ? 1u : 0u) << b;
        het |= (dubit == 2u ? 1u : 0u) << b;
    }
    uint32_t* o = planes + snp * 3u * ndw + d;
    o[0] = pres;
    o[ndw] = tot;
    o[2u * ndw] = het;
}

// One lane per SNP, phenotype column blockIdx.y. Yperm[p][L] is the permuted, zero-padded phenotype (permute_scores).
__global__ void __launch_bounds__(256) snp_score_kernel(const uint32_t* planes, uint64_t n_snps, uint32_t ndw, const float* Yperm,
                                                        uint32_t L, double mac, double* scores) {
    const uint64_t snp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t p = blockIdx.y;
    if (snp >= n_snps) return;
    const uint32_t* base = planes + snp * 3u * ndw;
    float dp[3];
    uint32_t cnt[3];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const uint32_t* w = base + pl * ndw;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t c = 0;
        for (uint32_t b = 0; b < ndw / 4u; b++) {
            const uint4 v = *reinterpret_cast<const uint4*>(w + 4u * b);
            const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
            c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
            const float* yb = Yperm + (size_t)p * L + 128u * b;
#pragma unroll
            for (int s = 0; s < 32; s++)
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    const int mk = ((int)(ww[l] << s)) >> 31;
                    acc[l] = acc[l] + __int_as_float(mk & __float_as_int(yb[4 * s + l]));
                }
        }
        dp[pl] = ((acc[0] + acc[1]) + acc[2]) + acc[3];  // sumsf[0] + sumsf[1] + sumsf[2] + sumsf[3], float (:62)
        cnt[pl] = c;
    }
    // per-SNP sums the reference accumulates while loading (:126-143); multiples of 1/4, exact in any order
    const double S_gi = (double)cnt[0] + 0.5 * (double)cnt[2];
    const double S_gi_2 = (double)cnt[0] + 0.25 * (double)cnt[2];
    const double N = (double)cnt[1];
    double out = 0.0;
    if (!((mac > S_gi) || (mac > (N - S_gi)))) {  // :157-158
        const double yigi = __dadd_rn((double)dp[0], __dmul_rn((double)dp[2], 0.5));
        const double score_sum = (double)dp[1];
        double r = __dsub_rn(__dmul_rn(N, yigi), __dmul_rn(S_gi, score_sum));
        r = __dmul_rn(r, r);
        const double den = __dmul_rn(N, __dsub_rn(__dmul_rn(N, S_gi_2), __dmul_rn(S_gi, S_gi)));
        out = r / den;
    }
    scores[(uint64_t)p * n_snps + snp] = out;
}

}  // namespace

hipError_t launch_snp_planes(const uint8_t* bed, uint64_t n_snps, uint32_t bytes_per_snp, const uint32_t* byte_idx,
                             const uint32_t* shift, uint32_t S, uint32_t ndw, uint32_t* planes, hipStream_t st) {
    const uint64_t n = n_snps * ndw;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(snp_planes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, bed, n_snps, bytes_per_snp, byte_idx,
                       shift, S, ndw, planes);
    return hipGetLastError();
}

hipError_t launch_snp_score(const uint32_t* planes, uint64_t n_snps, uint32_t ndw, const float* Yperm, uint32_t L, uint32_t n_pheno,
                            double mac, double* scores, hipStream_t st) {
    if (n_snps == 0 || n_pheno == 0) return hipSuccess;
    hipLaunchKernelGGL(snp_score_kernel, dim3((uint32_t)((n_snps + 255) / 256), n_pheno), dim3(256), 0, st, planes, n_snps, ndw, Yperm,
                       L, mac, scores);
    return hipGetLastError();
}

}  // namespace kgwas
