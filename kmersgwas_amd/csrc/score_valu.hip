// score_valu.hip — exact-order scorer on the vector ALU (gfx950).
//
// calculate_kmer_score (src/kmers_multiple_databases.cpp:327-363) with one lane per k-mer row and
// PC phenotype columns per thread. The reference's select-and-add is kept literally
// (acc += bit ? y : +0.0f, as an AND mask on y's bits), so non-finite phenotype values behave as
// they do on the CPU; y is wave-uniform and arrives through scalar loads. 1 + 2*PC lane-ops per
// table bit: the right tool for one or a few phenotype columns (where a 16-column MFMA tile would
// run mostly empty) and for columns holding inf/NaN (where 0*inf poisons the multiplicative form).
#include "score_common.h"

namespace kgwas {

template <int PC>
__global__ void __launch_bounds__(256) score_valu_kernel(ScoreArgs a) {
    extern __shared__ uint32_t lds_u32[];
    const uint32_t ndw = 2u * a.W_m;
    const uint32_t ldw = ndw + 1u;  // odd stride: lane == row reads are bank-conflict free
    const uint32_t TR = blockDim.x;
    const uint64_t row0 = (uint64_t)blockIdx.x * TR;

    // Row tile through LDS: HBM sees whole coalesced row segments, lanes then own one row each.
    for (uint32_t e = threadIdx.x; e < TR * ndw; e += TR) {
        const uint32_t rr = e / ndw, dw = e - rr * ndw;
        const uint64_t gr = row0 + rr;
        uint32_t v = 0;
        if (gr < a.n_rows && dw < a.src.avail_dw)
            v = a.src.base[gr * a.src.stride_dw + a.src.off_dw + dw] & a.dmask[dw];
        lds_u32[rr * ldw + dw] = v;
    }
    __syncthreads();

    const uint32_t rr = threadIdx.x;
    const uint64_t r = row0 + rr;
    const uint32_t* my = lds_u32 + rr * ldw;
    uint32_t n1 = 0;
    for (uint32_t dw = 0; dw < ndw; dw++) n1 += __popc(my[dw]);
    const bool in_range = r < a.n_rows;
    const bool pass = in_range && mac_pass(a, n1);

    const uint32_t p0 = blockIdx.y * PC;
    const uint32_t L = 64u * a.W_m;
    float acc[PC][4];
#pragma unroll
    for (int pc = 0; pc < PC; pc++)
#pragma unroll
        for (int l = 0; l < 4; l++) acc[pc][l] = 0.0f;

    const uint32_t nblk = a.W_m / 2u;
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t w[4];
#pragma unroll
        for (int l = 0; l < 4; l++) w[l] = my[4 * b + l];
        const float* yb = a.Yperm + (size_t)p0 * L + 128u * b;
#pragma unroll
        for (int s = 0; s < 32; s++) {
#pragma unroll
            for (int l = 0; l < 4; l++) {
                const int m = ((int)(w[l] << s)) >> 31;  // 0 or -1: bit 31-s of SSE lane l
#pragma unroll
                for (int pc = 0; pc < PC; pc++) {
                    const float y = yb[(size_t)pc * L + 4 * s + l];  // wave-uniform -> scalar load
                    acc[pc][l] = acc[pc][l] + __int_as_float(m & __float_as_int(y));
                }
            }
        }
    }

    if (blockIdx.y == 0 && in_range) {
        if (a.n1_out) a.n1_out[r] = n1;
        if (a.kmer_out) a.kmer_out[r] = a.file_rows[r * a.file_stride_w];
        if (pass && a.tested) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], 1ull);  // hipcc folds this into one atomic per wave
    }
    if (!in_range) return;
#pragma unroll
    for (int pc = 0; pc < PC; pc++) {
        const uint32_t p = p0 + pc;
        if (p >= a.n_pheno) break;
        const float yf = ((acc[pc][0] + acc[pc][1]) + acc[pc][2]) + acc[pc][3];  // :358, float adds
        finish_pair(a, r, p, yf, n1, pass, a.sums[p], a.thr ? a.thr[p] : 0.0);
    }
}

static hipError_t valu_dyn_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return hipSuccess;
}

hipError_t launch_score_valu(const ScoreArgs& a, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    const uint32_t ldw = 2u * a.W_m + 1u;
    uint32_t TR = 256;
    while (TR > 64 && (size_t)TR * ldw * 4u > 150u * 1024u) TR >>= 1;
    const size_t lds = (size_t)TR * ldw * 4u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    const uint32_t gx = (uint32_t)((a.n_rows + TR - 1) / TR);
    hipError_t e;
    if (a.n_pheno == 1) {
        if ((e = valu_dyn_lds((const void*)score_valu_kernel<1>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(score_valu_kernel<1>, dim3(gx, 1), dim3(TR), lds, st, a);
    } else {
        if ((e = valu_dyn_lds((const void*)score_valu_kernel<4>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(score_valu_kernel<4>, dim3(gx, (a.n_pheno + 3) / 4), dim3(TR), lds, st, a);
    }
    return hipGetLastError();
}

}  // namespace kgwas
