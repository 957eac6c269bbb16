// score_wide.hip — the int8 coarse filter (score_coarse.hip: same arithmetic, same bound, same survivors) for shapes
// whose one-slice operand set does not fit the LDS in one piece: BASELINE configs[3], 2048 samples x 201 columns, needs
// 13 operand tiles of 32 KB. coarse_kernel then walks four LDS groups of four tiles and expands every row four times
// (0.35 of the int8 peak in its steady launches, a fifth of the executed columns padding). Here
//  * ALL column tiles of a row tile stay in registers: a wave owns 64 rows x T tiles = 4 x T accumulators of 4 registers
//    (208 at T = 13), which takes the whole register file of a SIMD - one wave per SIMD, four per CU, accumulators in
//    the AccVGPR half (gfx950's unified 512-entry file);
//  * the operand tiles STREAM through LDS in stages of four MFMA steps (256 samples x T tiles = 4 T KB, 52 KB at
//    T = 13), double-buffered: stage s + 1 is copied global -> LDS by `global_load_lds_dwordx4` (no registers, 1 KB per
//    wave-instruction, the global image is the LDS image) while stage s is multiplied; one block barrier per stage;
//  * every row is loaded and expanded ONCE per pass, every operand fragment read from LDS feeds four row tiles
//    (T reads per 4 T MFMAs, as in coarse_kernel at its best shape), and per step the single wave of a SIMD issues
//    4 T independent MFMAs back to back with the next step's expansion and LDS reads between them.
// The block's 4 waves walk the stages in lockstep (each its own 64 rows), so the operand stream is read once per 256
// rows: 1.7 KB per row from L2 at T = 13 and 2048 samples.
#include <stdlib.h>

#include <type_traits>

#include "score_common.h"

namespace kgwas {

typedef int wi32x4 __attribute__((ext_vector_type(4)));

template <int T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
wide_kernel(CoarseArgs a, uint32_t rows_per_block, uint32_t n_rowblocks) {
    extern __shared__ wi32x4 wlds[];  // two stages of [4 steps][T][64] x 16 bytes, colc[3][T*16], exchange areas
    constexpr int RT = 4;
    constexpr int SLOTS = T * 16;
    constexpr uint32_t STAGE_VEC = 4u * T * 64u;  // i32x4 elements of a stage
    const uint32_t rb = blockIdx.x;
    if (rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t kg = lane >> 4, m = lane & 15u;
    const uint32_t n_stages = 2u * a.n_kgroups;
    float* colc = reinterpret_cast<float*>(wlds + 2u * STAGE_VEC);
    const int* colp = reinterpret_cast<const int*>(colc + 2 * SLOTS);
    float* wscr = colc + 3 * SLOTS + wave * 192u;  // wave-private: 64 x N1, 64 x (sqrt(d), E)
    if (threadIdx.x < SLOTS) {
        const CoarseCol cc = a.cols[threadIdx.x];
        float al = __builtin_huge_valf();  // padding / N1 column: nothing survives
        if (cc.pheno >= 0) al = (float)(sqrt(a.thr[cc.pheno]) * cc.kalpha);  // NaN threshold (frozen column) -> nothing survives
        colc[threadIdx.x] = al;
        colc[SLOTS + threadIdx.x] = cc.iu;
        reinterpret_cast<int*>(colc + 2 * SLOTS)[threadIdx.x] = cc.pheno;
    }
    // stage s of the operand stream -> LDS buffer `buf`: T wave-instructions of 1 KB per wave (4 waves x T = 4 T KB)
    const char* bq = reinterpret_cast<const char*>(a.Bq);
    auto fetch_stage = [&](uint32_t s, uint32_t buf) {
#pragma unroll
        for (int i = 0; i < T; i++) {
            const uint32_t c = (uint32_t)i * 4u + wave;  // KB of the stage
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bq + ((size_t)s * 4u * T + c) * 1024u + lane * 16u),
                                             (__attribute__((address_space(3))) void*)(wlds + buf * STAGE_VEC + c * 64u), 16, 0, 0);
        }
    };
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    uint32_t tested_local = 0;
    uint32_t sidx = 0;  // stages consumed so far: buffer parity
    fetch_stage(0u, 0u);

    uint32_t piece[RT][4], nxt[RT][4];
    uint32_t ro[RT];  // 32-bit byte offsets (launch_wide guarantees the chunk spans < 4 GiB) of the rows to fetch next
    auto set_rows = [&](uint64_t rb0) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            uint64_t r = rb0 + rt * 16u + m;
            if (r >= a.n_rows) r = a.n_rows - 1;
            ro[rt] = ((uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u;
        }
    };
    auto load_group = [&](uint32_t g, uint32_t (&dst)[RT][4]) {
        // bytes 64g + 16kg .. +15 of the row's bits; beyond the row's data the loads are clamped onto its last 8
        // bytes: whatever bits arrive there meet zero operands (sample slots >= S are zero in every column)
        uint32_t b0 = 64u * g + 16u * kg, b1 = b0 + 8u;
        b0 = b0 + 8u <= avail_b ? b0 : avail_b - 8u;
        b1 = b1 + 8u <= avail_b ? b1 : avail_b - 8u;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const uint2 lo = *reinterpret_cast<const uint2*>(rows_base + (ro[rt] + b0));
            const uint2 hi = *reinterpret_cast<const uint2*>(rows_base + (ro[rt] + b1));
            dst[rt][0] = lo.x;
            dst[rt][1] = lo.y;
            dst[rt][2] = hi.x;
            dst[rt][3] = hi.y;
        }
    };
    const uint32_t n_sets = (rows_per_block + 255u) / 256u;
    for (uint32_t ps = 0; ps < n_sets; ps++) {
        if (blk_row0 + (uint64_t)ps * 256u >= a.n_rows) break;  // block-uniform
        const uint64_t rbase = blk_row0 + (uint64_t)(ps * 4u + wave) * 64u;
        const bool live = rbase < a.n_rows;  // wave-uniform: a wave without rows still takes part in the operand stream
        wi32x4 acc[RT][T];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int t = 0; t < T; t++) acc[rt][t] = (wi32x4){0, 0, 0, 0};
        if (ps == 0) {  // the first pass-set's rows; later ones are fetched during the previous set's last stage
            set_rows(rbase);
            load_group(0u, piece);
        }
        for (uint32_t s = 0; s < n_stages; s++) {
            const uint32_t g = s >> 1, h = s & 1u;
            __builtin_amdgcn_s_waitcnt(0);  // this wave's share of stage s has landed in LDS (and the row bytes asked for a stage ago)
            __syncthreads();                // everybody's has; everybody is done with the other buffer
            const bool last_set = (ps + 1u == n_sets || blk_row0 + (uint64_t)(ps + 1u) * 256u >= a.n_rows);
            if (!(s + 1u == n_stages && last_set)) fetch_stage((s + 1u) % n_stages, (sidx + 1u) & 1u);
            if (h == 0u) {
                if (g) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int q = 0; q < 4; q++) piece[rt][q] = nxt[rt][q];
                }
            } else {
                // the NEXT sample group's row bytes (or group 0 of the wave's next 64 rows): a whole stage of MFMAs ahead
                if (g + 1u < a.n_kgroups) {
                    load_group(g + 1u, nxt);
                } else if (!last_set) {
                    set_rows(rbase + 256u);
                    load_group(0u, nxt);
                }
            }
            // Per step 4 T MFMAs, issued in order by the one wave of this SIMD: the NEXT step's operand fragments are
            // read from LDS between them (tile t's read follows tile t's MFMAs, into the other fragment buffer) and the
            // next step's expanded rows are made behind the first tile - pinned with sched_barrier, or the compiler moves
            // every read to just before its use and waits out the LDS latency once per tile. Only the stage's first
            // fragments are waited for (they cannot be asked for before the stage's barrier).
            const wi32x4* bg = wlds + (sidx & 1u) * STAGE_VEC + lane;
            wi32x4 Bb[2][T];
            wi32x4 Ac[RT], An[RT];
#pragma unroll
            for (int t = 0; t < T; t++) Bb[0][t] = bg[t * 64];
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                for (int q = 0; q < 4; q++) Ac[rt][q] = (int)((piece[rt][q] >> (4u * h)) & 0x01010101u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
#pragma unroll
                for (int t = 0; t < T; t++) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) acc[rt][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ac[rt], Bb[jj & 1][t], acc[rt][t], 0, 0, 0);
                    if (jj < 3) {
                        Bb[(jj + 1) & 1][t] = bg[((jj + 1) * T + t) * 64];
                        if (t == 0) {
#pragma unroll
                            for (int rt = 0; rt < RT; rt++)
#pragma unroll
                                for (int q = 0; q < 4; q++) An[rt][q] = (int)((piece[rt][q] >> (4u * h + (uint32_t)jj + 1u)) & 0x01010101u);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (jj < 3) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) Ac[rt] = An[rt];
                }
            }
            if (h == 1u && g + 1u == a.n_kgroups) {  // the pass's last stage: what was fetched belongs to the next pass-set
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int q = 0; q < 4; q++) piece[rt][q] = nxt[rt][q];
            }
            sidx++;
        }
        if (!live) continue;  // (wave-uniform; no barrier between here and the next stage's)

        // ---- per-row terms, tests, survivor emission: as coarse_kernel (score_coarse.hip), one slice ---------------
        float sqd[RT * 4], er[RT * 4];
        {
            int* n1s = reinterpret_cast<int*>(wscr);
            float2* trm = reinterpret_cast<float2*>(wscr + 64);
            if (m == 15u) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) *reinterpret_cast<wi32x4*>(n1s + kg * 16u + rt * 4) = acc[rt][T - 1];
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t n1r = (uint32_t)n1s[lane];  // row slot m of this kg
            const uint64_t left = a.n_rows - rbase;
            const uint32_t rows_here = left < 64u ? (uint32_t)left : 64u;
            const bool mac_any = a.S >= 2u * a.min_count;
            const uint32_t span = a.S - 2u * a.min_count;
            const bool ok = mac_any & ((m >> 2) * 16u + kg * 4u + (m & 3u) < rows_here) & ((n1r - a.min_count) <= span);
            tested_local += ok ? 1u : 0u;
            const float f = (float)n1r;
            const float sq = __builtin_amdgcn_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; 1 ulp sqrt; (1 - 2^-20)
            float2 tm;
            tm.x = ok ? sq : __builtin_huge_valf();
            tm.y = (a.eg_max + fminf(a.rall_max, f * a.rmax_max)) * 1.000001f;
            trm[lane] = tm;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < RT * 4; i += 2) {
                const float4 v = *reinterpret_cast<const float4*>(trm + kg * 16u + i);
                sqd[i] = v.x;
                er[i] = v.y;
                sqd[i + 1] = v.z;
                er[i + 1] = v.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        auto pair_margin = [&](int i, int g, float al) {
            return fmaf(-al, sqd[i], fabsf((float)acc[i >> 2][g][i & 3]));  // NaN (frozen column) never wins a maximum
        };
        float alc[T];
#pragma unroll
        for (int g = 0; g < T; g++) alc[g] = colc[g * 16 + m];
        uint64_t hit[RT * 4];
        {
            float mx[RT * 4];
#pragma unroll
            for (int i = 0; i < RT * 4; i++) mx[i] = -__builtin_huge_valf();
#pragma unroll
            for (int g = 0; g < T; g++)
#pragma unroll
                for (int i = 0; i < RT * 4; i++) mx[i] = fmaxf(mx[i], pair_margin(i, g, alc[g]));
#pragma unroll
            for (int i = 0; i < RT * 4; i++) hit[i] = __ballot(mx[i] + er[i] >= 0.0f);
        }
        uint64_t hit_any = 0;
#pragma unroll
        for (int i = 0; i < RT * 4; i++) hit_any |= hit[i];
        if (hit_any) {  // wave-uniform
            // mb[g] bit i = pair (row slot i, column g*16 + m) survives; rebuilt for the row slots that had a hit
            uint32_t mb[T];
#pragma unroll
            for (int g = 0; g < T; g++) mb[g] = 0;
#pragma unroll
            for (int i = 0; i < RT * 4; i++) {
                if (hit[i]) {  // wave-uniform
#pragma unroll
                    for (int g = 0; g < T; g++) {
                        float al = alc[g];
                        asm volatile("" : "+v"(al));
                        mb[g] |= (pair_margin(i, g, al) + er[i] >= 0.0f) ? (1u << i) : 0u;
                    }
                }
            }
            // survivors as quarter words of the bitmap (see coarse_kernel; nibble-transposed words)
            unsigned short* bm16 = reinterpret_cast<unsigned short*>(a.bitmap) + (rbase >> 6) * 4u + kg;
#pragma unroll
            for (int g = 0; g < T; g++)
                if (mb[g]) bm16[(uint64_t)colp[g * 16 + m] * a.words_per_col * 4u] = (unsigned short)mb[g];
        }
    }
    if (a.tested) {
        uint32_t v = tested_local;  // every lane counted one row per pass
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

size_t wide_lds_bytes(uint32_t T) { return (size_t)2u * 4u * T * 1024u + 3u * T * 16u * 4u + 4u * 768u; }

template <int T>
static hipError_t launch_wide_t(const CoarseArgs& a, uint32_t rows_per_block, uint32_t n_rowblocks, size_t lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void*)wide_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((wide_kernel<T>), dim3(n_rowblocks), dim3(256), lds, st, a, rows_per_block, n_rowblocks);
    return hipGetLastError();
}

hipError_t launch_wide(const CoarseArgs& a, uint32_t T, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    if (a.n_slices != 1 || a.n_lgroups != 1) return hipErrorInvalidValue;
    const size_t lds = wide_lds_bytes(T);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if ((a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit byte offsets
    rows_per_block = (rows_per_block + 255u) / 256u * 256u;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
    switch (T) {
        case 9: return launch_wide_t<9>(a, rows_per_block, n_rowblocks, lds, st);
        case 10: return launch_wide_t<10>(a, rows_per_block, n_rowblocks, lds, st);
        case 11: return launch_wide_t<11>(a, rows_per_block, n_rowblocks, lds, st);
        case 12: return launch_wide_t<12>(a, rows_per_block, n_rowblocks, lds, st);
        case 13: return launch_wide_t<13>(a, rows_per_block, n_rowblocks, lds, st);
        case 14: return launch_wide_t<14>(a, rows_per_block, n_rowblocks, lds, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace kgwas
