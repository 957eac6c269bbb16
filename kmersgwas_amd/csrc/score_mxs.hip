// score_mxs.hip — the block-scaled filter of score_mx.hip in its OPERAND-STREAMING form, for shapes whose slice operands
// do not fit the LDS in one piece (2048 samples x 201 columns: 520 KB; 1135 x 101: 161 KB). Same contract, same operand
// bytes, same accumulators, same test and the same survivors' bitmap as mx_kernel (see there for the sample <-> k maps,
// the FP4 / FP6 slice grids and the bound; calculate_kmer_score: src/kmers_multiple_databases.cpp:327-363) - what changes
// is where the operands are while a row is multiplied:
//
//   mx_kernel   all steps of CT column tiles resident in LDS; a shape with more columns than fit becomes several "LDS
//               groups", and EVERY row is loaded, expanded to FP4 operands and tested once per group (five times at
//               2048 x 201, 15 column tiles for the 12.6 the columns need, >= 1.86 x the algorithmic HBM traffic).
//   mxs_kernel  the waves of a block keep the accumulators of ALL column tiles of an operand group (up to 14) for their row tiles
//               and walk the sample axis ONCE: the operands of one step (128 samples x all column tiles: 2560 B per tile, 35 KB
//               for 14) stream through a ring of LDS slots, fetched with `global_load_lds_dwordx4` (global -> LDS, no registers)
//               three steps ahead of their use by all waves of the block together, one `s_barrier` per step. Every row is
//               loaded and expanded once; the operands (L2-resident: 560 KB per operand group) cross the L2 -> LDS path once
//               per block pass (256 rows). No limit on the number of samples: nothing but one step's operands is in LDS.
//
//   ring protocol, step s (block-wide step counter, continuing across the block's passes; RING slots, AHEAD = RING - 1):
//     top of step s:  s_waitcnt vmcnt(N)   this wave's part of slab s + 1 (issued in step s + 1 - AHEAD) has landed - N = the
//                                          constant number of younger operations (see WAIT_N): they stay in flight
//                     s_barrier            (B_s) every wave's part has; every wave is through step s - 1's operand reads
//                     issue slab s + AHEAD into slot (s + AHEAD) % RING = the slot step s - 1 read; then RT row requests / fillers
//     during step s:  operand reads of step s (slot s % RING) and, in its last unit, of step s + 1's first unit
//   Every vector-memory operation of the main loop - the slabs, the row pieces (LDS-DMA into a per-wave staging area, picked up
//   with ds_read_b128), the fillers - is issued by inline assembly: the compiler would otherwise wait for every LDS-DMA in
//   flight before the next ds_read that may alias it, and for ALL outstanding operations where it first uses a loaded value.
//   It neither sees these transfers nor counts them, so the waits above are explicit, and every such statement carries a
//   "memory" clobber (the ring changes behind the compiler's back).
//
// Shapes (template): CT column tiles per wave, RT row tiles per wave, NG column groups among the waves of a block (a group =
// TH / 64 / NG waves with their own rows; the groups of a block share rows and split the column tiles, each with its own ones
// column), TH threads, RING slots (4 where they fit the LDS, else 3).
//   < 7, 4, 2, 512>  2048 x 201 (the default above 7 tiles): waves w and w + 4 work on the same 64 rows, 7 tiles each (14 for 12.6)
//   < 7, 4, 1, 512>  up to 7 tiles (1135 x 101: 9 steps x 7 tiles instead of two LDS groups of 4): eight waves of 64 rows
//   <13, 2, 1, 512>  KGWAS_MXS_FORM=1: 13 tiles, eight waves of 32 rows (104 accumulator registers; twice the LDS reads per MFMA)
//   <13, 4, 1, 256>  KGWAS_MXS_FORM=2: one wave per SIMD with the 512-register budget, 64 rows x 13 tiles (208)
//
// Measured (round 5): at 2048 x 201 and 1135 x 101 the default shapes run within +-2 % of the resident plan's
// launches (40.4 against 40.3 ms, 13.3 against 13.4 ms per 100 M rows) with the rows read from HBM once instead of once per LDS
// group; the 13-tile shapes are slower (46.7 / 46.8 ms). What the form adds is reach: 6000 accessions x 101 columns go through
// the filter (31.7 ms per 20 M rows; the exact scorer, the only path such a panel had: 4.7 s).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "score_common.h"

#ifndef KGWAS_MXS_DEPHASE  // 1: the upper half of a block's waves runs its passes half a pass behind the lower half (see `dephase`): measured SLOWER, off
#define KGWAS_MXS_DEPHASE 0
#endif
#ifndef KGWAS_MXS_ROWS_NT
#define KGWAS_MXS_ROWS_NT 0  // (measured: 3.7 x the algorithmic HBM bytes - the second column group's request of a row then always misses - and + 10 % time)
#endif
#ifndef KGWAS_MXS_ABLATE  // timing experiments only (wrong results): 1 no tests, 8 no row loads, 16 no operand DMA, 64 no barriers
#define KGWAS_MXS_ABLATE 0
#endif

#ifndef KGWAS_MXS_PROF  // experiments: cycle counters of wave 0 of every block (sync waits, steps, epilogues) summed into mxs_prof[]
#define KGWAS_MXS_PROF 0
#endif

namespace kgwas {

#if KGWAS_MXS_PROF
__device__ unsigned long long mxs_prof[8];  // [0] cycles in step_sync, [1] cycles in steps (incl. sync), [2] epilogue cycles, [3] steps, [4] passes, [5] whole-kernel cycles of wave 0
extern "C" int kgwas_debug_mxs_prof(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mxs_prof), sizeof(mxs_prof)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mxs_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

typedef int mxsv8i __attribute__((ext_vector_type(8)));
typedef float mxsv4f __attribute__((ext_vector_type(4)));

namespace {
constexpr uint32_t MXS_SB = 2560u;     // FP6 slice (64 lanes x 16 B, then 64 lanes x 8 B) + FP4 slice (64 lanes x 16 B) of one (step, column tile)
constexpr uint32_t MXS_PART0 = 1536u;
}  // namespace

template <int CT, int RT, int NG, int TH, int RING>
__global__ void __launch_bounds__(TH) mxs_kernel(MxArgs a, uint32_t rows_per_block, uint32_t n_rowblocks) {
    extern __shared__ uint4 mslds[];  // ring[3][CTA x 2560], colc[3][SLOTS_ALL] (alpha, -, column index), per-wave row-term exchange
    constexpr uint32_t MXS_RING = RING;        // ring slots
    constexpr uint32_t MXS_AHEAD = RING - 1u;  // a slab is requested this many steps before the step that multiplies with it
    constexpr int CTA = CT * NG;          // column tiles of the block's operand slab
    constexpr int SLOTS = CT * 16;        // operand columns of a column group
    constexpr int SLOTS_ALL = CTA * 16;
    constexpr int WPG = TH / 64 / NG;     // waves per column group
    constexpr int NSL = RT * 4;           // row slots per lane
    constexpr uint32_t SLAB = CTA * MXS_SB;
    constexpr uint32_t SLOT = (SLAB + 1023u) / 1024u * 1024u;  // ring slot (the last transfer of a slab may run up to 1 KB past it)
    static_assert(RT == 2 || RT == 4, "row tiles per wave");
    uint32_t rb = blockIdx.x, lg = 0;
    if (a.n_lgroups > 1) {  // every (row block, operand group) pair is a block; the groups of a row block run next to each other on one XCD
        const uint32_t idx = blockIdx.x >> 3;
        rb = (idx / a.n_lgroups) * 8u + (blockIdx.x & 7u);
        lg = idx % a.n_lgroups;
    }
    if (rb >= n_rowblocks) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t grp = wave / WPG, wig = wave % WPG;
    const uint32_t kb = lane >> 4, m = lane & 15u;
    const uint32_t n_steps = 4u * a.n_full + a.n_quarter;
    char* lds = reinterpret_cast<char*>(mslds);
    float* colc = reinterpret_cast<float*>(lds + MXS_RING * SLOT);
    const int* colp = reinterpret_cast<const int*>(colc + 2 * SLOTS_ALL) + grp * SLOTS;
    float* wscr = colc + 3 * SLOTS_ALL + wave * (RT * 48u);  // wave-private: RT*16 x N1, RT*16 x (sqrt(d), E)
    const uint32_t rows_per_pass = WPG * (RT * 16u);
    const uint64_t blk_row0 = (uint64_t)rb * rows_per_block;
    const float Nf = (float)a.S;
    const char* rows_base = reinterpret_cast<const char*>(a.src.base);
    const uint32_t avail_b = a.src.avail_dw * 4u;
    const int sc0 = (int)a.scale0;
    uint32_t tested_local = 0;

    // passes of this block: the same for all its waves (they meet at a barrier every step)
    uint32_t n_passes;
    {
        const uint64_t left = a.n_rows - blk_row0;
        const uint32_t rows = left < rows_per_block ? (uint32_t)left : rows_per_block;
        n_passes = (rows + rows_per_pass - 1u) / rows_per_pass;
    }

    // ---- the operand stream: slab (step) -> ring slot, 16-byte units, unit u = i * TH + thread of round i
    const uint8_t* slab_src = a.Bq + (size_t)lg * n_steps * SLAB;
    const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    constexpr uint32_t N_UNITS = SLAB / 16u;
    constexpr uint32_t ROUNDS = (N_UNITS + TH - 1u) / TH;
    const uint32_t voff = lane * 16u;
    uint32_t dma_step = 0, dma_slot = 0;  // the next slab to fetch and where it goes
    // No branches in here (a branch in the step splits the main loop's body, see mx_kernel): a wave whose units of the last
    // round lie beyond the slab fetches its units of round 0 again, and the one wave that straddles the slab's end reads on
    // into the next slab (the buffer ends with 1 KB of padding) and writes into the slot's own padding (SLOT is SLAB rounded
    // up to whole KB).
    auto issue_slab = [&]() {
        if (KGWAS_MXS_ABLATE & 16) return;
        const uint8_t* src = slab_src + (size_t)dma_step * SLAB;
        const uint32_t dst = lds_ring + dma_slot * SLOT;
#pragma unroll
        for (uint32_t i = 0; i < ROUNDS; i++) {
            uint32_t ub = i * TH + wave * 64u;  // first unit of this wave's instruction
            if ((i + 1u) * TH > N_UNITS) ub = ub < N_UNITS ? ub : wave * 64u;  // (a scalar select)
            const uint32_t m0v = dst + ub * 16u;
            const uint8_t* sb = src + (size_t)ub * 16u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sb) : "memory");
        }
        dma_step = dma_step + 1u == n_steps ? 0u : dma_step + 1u;
        dma_slot = dma_slot + 1u == MXS_RING ? 0u : dma_slot + 1u;
    };
    // The vector-memory sequence of EVERY step is [ROUNDS slab transfers][RT row operations] - the row operations being the
    // wave's row loads of the next 512-sample group in the one step per group that issues them, and RT four-byte LDS-DMA
    // fillers (a hot line into a scratch area nobody reads) in every other step. Loads complete in order and `s_waitcnt vmcnt`
    // takes an immediate: with a uniform sequence "slab s + 1 has landed" is the constant vmcnt(WAIT_N) at the top of step s,
    // however many younger operations - the next slabs, the row loads - are still in flight. A row load issued in step s must
    // then have landed by the top of step s + MXS_AHEAD (it is older than the slab requested in step s + 1, which is waited
    // for there): three steps, ~7000 cycles, against ~4000 of HBM latency. (With vmcnt(0) every step - the first version -
    // every wave waited out its row loads' HBM latency one step after issuing them, and the block with it: +13 % kernel time.)
    constexpr uint32_t WAIT_N = RT + (MXS_AHEAD - 2u) * (ROUNDS + RT);
    static_assert(MXS_AHEAD >= 2u && WAIT_N < 64u, "vmcnt is a 6-bit immediate");
    const uint32_t lds_fill = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + MXS_RING * SLOT + 3u * SLOTS_ALL * 4u + (TH / 64u) * (RT * 48u * 4u) + wave * 256u;
    // this wave's row staging area: RT x (64 lanes x 16 B), behind the fillers' scratch (one per wave: waves of different column
    // groups work on the same rows, but - KGWAS_MXS_DEPHASE - half a pass apart)
    const uint32_t stage_off = MXS_RING * SLOT + 3u * SLOTS_ALL * 4u + (TH / 64u) * (RT * 48u * 4u) + (TH / 64u) * 256u + (KGWAS_MXS_DEPHASE ? wave : wig) * (RT * 1024u);
    const uint32_t lds_stage = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + stage_off;
    const uint32_t voff4 = lane * 4u;
    auto filler_ops = [&]() {
        if (KGWAS_MXS_ABLATE & 16) return;
#pragma unroll
        for (int i = 0; i < RT; i++)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_fill), "v"(voff4), "s"(slab_src) : "memory");
    };
#if KGWAS_MXS_PROF
    unsigned long long pf_sync = 0, pf_step = 0, pf_epi = 0, pf_nstep = 0, pf_bar = 0, pf_issue = 0;
    const unsigned long long pf_k0 = __builtin_readcyclecounter();
#endif
    // top of step s: this wave's part of slab s + 1 has landed; B_s (every wave's part has, and every wave is through step
    // s - 1's operand reads); slab s + MXS_AHEAD goes into the slot step s - 1 read. Past the block's last step the stream simply
    // wraps around: fetched, never read, waited for before the kernel ends.
    auto step_sync = [&]() {
#if KGWAS_MXS_PROF
        const unsigned long long t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_N) : "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (!(KGWAS_MXS_ABLATE & 64)) asm volatile("s_barrier" ::: "memory");
        pf_bar += __builtin_readcyclecounter() - t1;
        pf_sync += __builtin_readcyclecounter() - t0;
#else
        if (KGWAS_MXS_ABLATE & 64)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_N) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WAIT_N) : "memory");
#endif
    };

    {
        if (threadIdx.x < SLOTS_ALL) {
            const CoarseCol cc = a.cols[lg * SLOTS_ALL + threadIdx.x];
            float al = __builtin_huge_valf();  // padding / N1 column: nothing survives
            if (cc.pheno >= 0) al = (float)(sqrt(a.thr[cc.pheno]) * cc.kalpha);  // NaN threshold (frozen column) -> NaN -> nothing survives
            colc[threadIdx.x] = al;
            colc[SLOTS_ALL + threadIdx.x] = cc.iu;
            reinterpret_cast<int*>(colc + 2 * SLOTS_ALL)[threadIdx.x] = cc.pheno;
        }
#pragma unroll
        for (uint32_t i = 0; i < MXS_AHEAD; i++) issue_slab();  // slabs 0 .. MXS_AHEAD - 1
    }
    __syncthreads();  // (colc; the compiler's own barrier with its waits)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the first slabs are in the ring

    // (32-bit byte offsets of a lane's rows: launch_mxs guarantees the chunk spans < 4 GiB)
    const uint64_t wave_row0 = blk_row0 + wig * (RT * 16u);
    auto set_rows = [&](uint32_t (&o)[RT], uint64_t rb0) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            uint64_t r = rb0 + rt * 16u + m;
            if (KGWAS_MXS_ABLATE & 128) r &= 4095u;  // (experiments: every row load hits the L2)
            if (r >= a.n_rows) r = a.n_rows - 1;
            o[rt] = ((uint32_t)r * (uint32_t)a.src.stride_dw + a.src.off_dw) * 4u;
        }
    };
    // This lane's 16 bytes of full group g of each of its RT rows (the four kb-lanes of a row fetch the group's 64 bytes as one
    // contiguous piece; a full group's bytes always exist in the row: n_full = S / 512), requested as LDS-DMA into the wave's
    // staging area - lane l's bytes land at 16 l - and picked up with one ds_read_b128 per row tile three steps later. Through
    // LDS and not into registers: every vector-memory operation of the main loop is then one the compiler neither sees nor
    // waits for (a compiler-visible load made it drain the whole queue - slabs in flight included - where it first used the
    // data), and no register is the target of a load the compiler does not know about.
    auto request_group = [&](const uint32_t (&o)[RT], uint32_t g) {
        const uint32_t b0 = 64u * g + 16u * kb;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            uint32_t off = o[rt] + b0;
            if (KGWAS_MXS_ABLATE & 8) off = lane * 16u;
            const uint32_t m0v = lds_stage + rt * 1024u;
            // (non-temporal: a row is read once - by both column groups' waves at nearly the same time -, and must not push the
            // operand slabs, which every block re-reads once per pass, out of the L2)
#if KGWAS_MXS_ROWS_NT
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(m0v), "v"(off), "s"(rows_base) : "memory");
#else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(off), "s"(rows_base) : "memory");
#endif
        }
    };
    auto pickup_group = [&](uint32_t (&pc)[RT][4]) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const uint4 v = *reinterpret_cast<const uint4*>(lds + stage_off + rt * 1024u + lane * 16u);
            pc[rt][0] = v.x;
            pc[rt][1] = v.y;
            pc[rt][2] = v.z;
            pc[rt][3] = v.w;
        }
    };
    // Slice operands of the column tiles, read from the ring one UNIT (two column tiles) ahead of their MFMAs, as in mx_kernel.
    constexpr int UT = 2;
    constexpr int NU = (CT + UT - 1) / UT;
    static_assert(NU >= 2, "at least three column tiles per wave");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x4 Bx[CT], Bz[CT];  // dwords 0-3 of the first slice's operand, the second slice's
    u32x2 By[CT];          // dwords 4-5 of the first slice's
#pragma unroll
    for (int t = 0; t < CT; t++) Bx[t] = Bz[t] = (u32x4){0u, 0u, 0u, 0u}, By[t] = (u32x2){0u, 0u};
    struct StepAddr {
        uint32_t o16, o8, o8b;
    };
    // operand addresses of the step whose slab lies in ring slot `slot`
    auto step_addr = [&](uint32_t slot) {
        StepAddr sa;
        const uint32_t base = slot * SLOT + grp * (CT * MXS_SB);
        sa.o16 = base + lane * 16u;
        sa.o8 = base + lane * 8u + 1024u;
        sa.o8b = sa.o8 + MXS_SB;
        asm volatile("" : "+v"(sa.o16), "+v"(sa.o8), "+v"(sa.o8b));
        return sa;
    };
    auto read_unit = [&](int u, const StepAddr& sa) {
#pragma unroll
        for (int t = UT * u; t < (UT * u + UT < CT ? UT * u + UT : CT); t++) {
            const char* p16 = lds + sa.o16 + t * MXS_SB;
            const char* p8 = (t & 1) ? lds + sa.o8b + (t - 1) * MXS_SB : lds + sa.o8 + t * MXS_SB;
            Bx[t] = *reinterpret_cast<const u32x4*>(p16);
            By[t] = *reinterpret_cast<const u32x2*>(p8);
            Bz[t] = *reinterpret_cast<const u32x4*>(p16 + MXS_PART0);
        }
    };
    uint32_t slot_cur = 0;  // ring slot of the step about to run
    // Row pieces: `piece` = this lane's 16 bytes of the CURRENT 512-sample group of each of its RT rows. The group after it (the
    // next group of the pass or group 0 of the wave's next rows) is requested in step 1 of the current group and picked up at
    // the top of its own step 0, behind that step's wait.
    uint32_t piece[RT][4];
    // The waves w and w + 4 of a block share a SIMD and meet at every step's barrier. Had both the same pass boundaries, both
    // would run their epilogues - 4000 cycles of vector work per 43 000-cycle pass - at the same time, with nobody feeding the
    // matrix pipe (in the free-running resident kernel the two waves' epilogues hide behind each other's MFMAs). The slab stream
    // is cyclic and the accumulation order is free, so the upper half of the block's waves (LATE) takes its passes HALF A PASS
    // behind: a late wave's pass is the groups off .. n_full - 1, the quarter steps, then the groups 0 .. off - 1 of the next
    // cycle of the stream (off = n_full / 2 whole groups) - the same slabs at the same barriers, its epilogue in the middle of
    // the early waves' main loop and vice versa. The late waves idle through the block's first 4 off steps (barrier and
    // transfers only), the early ones through the 4 off steps behind their last pass.
    // MEASURED (round 5, alternating builds on one box, ms of filter per 100 M rows): 2048 x 201 45.7-46.3 with the offset
    // against 42.3 without, 1135 x 101 14.3-14.4 against 13.0 - slower by 8-10 %, although the results are identical and the
    // epilogues provably no longer coincide: partner waves then fetch their rows at different times (the second fetch no
    // longer hits what the first brought in) and every block runs 4 off steps more. Compiled out (KGWAS_MXS_DEPHASE=0).
    const bool dephase = KGWAS_MXS_DEPHASE && TH >= 512 && a.n_full >= 2u;
    const bool late = dephase && wave >= TH / 128;
    const uint32_t off = late ? a.n_full / 2u : 0u;
    const uint32_t n_dummy = dephase ? 4u * (a.n_full / 2u) : 0u;
    uint32_t pf_ps = 0, pf_g = off;  // the group the next request fetches (in the wave's own order: off, off + 1, ..., off - 1)
    auto prefetch = [&]() {
        uint32_t o[RT];
        set_rows(o, wave_row0 + (uint64_t)pf_ps * rows_per_pass);
        request_group(o, pf_g);
        pf_g = pf_g + 1u == a.n_full ? 0u : pf_g + 1u;
        pf_ps = pf_g == off ? pf_ps + 1u : pf_ps;
    };
    if (a.n_full) prefetch();  // (the first group: picked up at the top of the first step, as every group's pieces are)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the steps' waits count from an empty queue)
    StepAddr sadr = step_addr(0);
    read_unit(0, sadr);  // step 0 of the first pass; every step's last unit fetches the next step's first
    // a step without work for this wave: the barrier and the wave's share of the transfers (the sequence of vector-memory
    // operations stays uniform), and the next step's first operands
    auto dummy_step = [&]() __attribute__((always_inline)) {
        step_sync();
        issue_slab();
        filler_ops();
        const uint32_t slot_next = slot_cur + 1u == MXS_RING ? 0u : slot_cur + 1u;
        sadr = step_addr(slot_next);
        read_unit(0, sadr);
        slot_cur = slot_next;
    };
    if (late)
        for (uint32_t i = 0; i < n_dummy; i++) dummy_step();
    for (uint32_t ps = 0; ps < n_passes; ps++) {
        const uint64_t rbase = wave_row0 + (uint64_t)ps * rows_per_pass;
        mxsv4f acc[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int t = 0; t < CT; t++) acc[rt][t] = (mxsv4f){0.0f, 0.0f, 0.0f, 0.0f};

        auto mfma_unit = [&](int u, const mxsv8i (&A)[RT], int sa) {
#pragma unroll
            for (int t = UT * u; t < (UT * u + UT < CT ? UT * u + UT : CT); t++) {
                const mxsv8i B0 = {(int)Bx[t].x, (int)Bx[t].y, (int)Bx[t].z, (int)Bx[t].w, (int)By[t].x, (int)By[t].y, 0, 0};
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    acc[rt][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[rt], B0, acc[rt][t], 4, 2, 0, sa, 0, sc0);
                const mxsv8i B1 = {(int)Bz[t].x, (int)Bz[t].y, (int)Bz[t].z, (int)Bz[t].w, 0, 0, 0, 0};
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    acc[rt][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[rt], B1, acc[rt][t], 4, 4, 0, sa, 0, 0x7F7F7F7F);
            }
        };
        // step = [B_s, slab s + AHEAD, row operations | operands A | reads of unit 1 | MFMAs of unit 0 | reads of unit 2 | MFMAs of unit 1
        //         | ... | reads of the next step's unit 0 | MFMAs of the last unit]
        // kind: 0..3 = step j of a 512-sample group (table bits from `piece`), 4 = a quarter step (operands handed in)
        auto run_step = [&](auto kind, mxsv8i (&A)[RT], int sa) __attribute__((always_inline)) {
            constexpr int J = decltype(kind)::value;
            __builtin_amdgcn_sched_barrier(0);
#if KGWAS_MXS_PROF
            const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
            step_sync();
            auto vmem_ops = [&]() {  // this step's vector-memory operations: the slab MXS_AHEAD steps ahead, then RT row requests or fillers
#if KGWAS_MXS_PROF
                const unsigned long long t2 = __builtin_readcyclecounter();
#endif
                issue_slab();
                // (two column groups share rows and staging areas and BOTH request them - identical transfers, the second one an L2
                // hit. Letting only the lower group request - the upper one finds the rows behind the barrier - measured 4 % slower
                // with the same HBM-side traffic; non-temporal requests 10 % slower with 3.7 x the algorithmic traffic.)
                if (J == 1)
                    prefetch();
                else
                    filler_ops();
#if KGWAS_MXS_PROF
                pf_issue += __builtin_readcyclecounter() - t2;
#endif
            };
            vmem_ops();
            __builtin_amdgcn_sched_barrier(0);
            if (J == 0) pickup_group(piece);
            if (J < 4) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    A[rt] = (mxsv8i){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        A[rt][q] = J < 3 ? (int)(piece[rt][q] & (0x11111111u << J)) : (int)((piece[rt][q] >> 1) & 0x44444444u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t slot_next = slot_cur + 1u == MXS_RING ? 0u : slot_cur + 1u;
            const StepAddr nadr = step_addr(slot_next);
#pragma unroll
            for (int u = 0; u < NU; u++) {
                if (u + 1 < NU)
                    read_unit(u + 1, sadr);
                else
                    read_unit(0, nadr);
                __builtin_amdgcn_sched_barrier(0);
                mfma_unit(u, A, sa);
                __builtin_amdgcn_sched_barrier(0);
            }
            sadr = nadr;
            slot_cur = slot_next;
#if KGWAS_MXS_PROF
            pf_step += __builtin_readcyclecounter() - ts0;
            pf_nstep++;
#endif
        };

        __builtin_amdgcn_s_setprio(0);
        auto group_steps = [&]() __attribute__((always_inline)) {
            mxsv8i A[RT];
            // nibble bit 0 / 1 / 2 / 2 (bit 3 shifted down): 0.5 / 1.0 / 2.0 / 2.0 x 2^0 / 2^-1 / 2^-2 / 2^-2 = 0.5
            run_step(std::integral_constant<int, 0>(), A, 0x7F7F7F7F);
            run_step(std::integral_constant<int, 1>(), A, 0x7E7E7E7E);
            run_step(std::integral_constant<int, 2>(), A, 0x7D7D7D7D);
            run_step(std::integral_constant<int, 3>(), A, 0x7D7D7D7D);
        };
        for (uint32_t g = off; g < a.n_full; g++) group_steps();
        // quarter groups: 128 samples per step, the lane's own dword shifted by 0..3
        uint32_t ro[RT];  // (made here, not kept across the groups' steps: eight registers the main loop has no room for)
        if (a.n_quarter) set_rows(ro, rbase);
        for (uint32_t x = 0; x < a.n_quarter; x++) {
            uint32_t b0 = 64u * a.n_full + 16u * x + 4u * kb;
            b0 = b0 + 4u <= avail_b ? b0 : avail_b - 4u;
            mxsv8i A[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const uint32_t w = (KGWAS_MXS_ABLATE & 8) ? ro[rt] + b0 : *reinterpret_cast<const uint32_t*>(rows_base + (ro[rt] + b0));
                A[rt] = (mxsv8i){(int)(w & 0x11111111u), (int)((w >> 1) & 0x11111111u), (int)((w >> 2) & 0x11111111u),
                                 (int)((w >> 3) & 0x11111111u), 0, 0, 0, 0};
            }
            run_step(std::integral_constant<int, 4>(), A, 0x7F7F7F7F);
        }
        for (uint32_t g = 0; g < off; g++) group_steps();  // (late waves: the first groups, from the stream's next cycle)

#if KGWAS_MXS_PROF
        const unsigned long long te0 = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_setprio(3);  // the epilogue at raised priority (mx_kernel)
        // Per-row terms, exactly as in mx_kernel: N1 from the ones column (slot 15 of the group's last column tile), sqrt(d)
        // rounded down (+inf for a row that does not exist or fails the MAC filter), E(N1) rounded up; one row per lane through a
        // wave-private exchange area.
        float2* trm = reinterpret_cast<float2*>(wscr + RT * 16);
        {
            float* n1s = wscr;
            if (m == 15u) {
#pragma unroll
                for (int rt = 0; rt < RT; rt++) *reinterpret_cast<mxsv4f*>(n1s + kb * NSL + rt * 4) = acc[rt][CT - 1];
            }
            __builtin_amdgcn_wave_barrier();
            const uint64_t left = rbase < a.n_rows ? a.n_rows - rbase : 0;
            const uint32_t rows_here = left < RT * 16u ? (uint32_t)left : RT * 16u;
            const bool mac_any = a.S >= 2u * a.min_count;  // else no N1 can satisfy mc <= N1 <= S - mc
            const uint32_t span = a.S - 2u * a.min_count;
            if (lane < RT * 16u) {
                // entry e = lane = (kb' = e / NSL, slot i = e % NSL): row 16 (i / 4) + 4 kb' + i % 4
                const uint32_t e = lane;
                const uint32_t kbe = e / NSL, ie = e % NSL;
                const uint32_t row = 16u * (ie >> 2) + 4u * kbe + (ie & 3u);
                const float f = n1s[e];
                const uint32_t n1r = (uint32_t)f;
                const bool ok = mac_any & (row < rows_here) & ((n1r - a.min_count) <= span);
                if (lg == 0 && grp == 0) tested_local += ok ? 1u : 0u;
                const float sq = __builtin_amdgcn_sqrtf(f * (Nf - f)) * 0.99999905f;  // d < 2^24 is exact; 1 ulp sqrt; (1 - 2^-20)
                float2 tm;
                tm.x = ok ? sq : __builtin_huge_valf();
                tm.y = (a.eg_max + fminf(a.rall_max, f * a.rmax_max)) * 1.000001f;
                trm[e] = tm;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // The test (mx_kernel): a running maximum of |acc| per row slot, ONE fma + compare per row slot with the smallest
        // alpha of the lane's columns, the per-pair test only for row slots that hit.
        float alc[CT];
        float al_min = __builtin_huge_valf();
#pragma unroll
        for (int t = 0; t < CT; t++) {
            alc[t] = colc[grp * SLOTS + t * 16 + m];
            al_min = fminf(al_min, alc[t]);  // (a NaN alpha - frozen column - is skipped; +inf = padding / ones column)
        }
        const bool ones_lane = m == 15u;  // slot 15 of the last tile is the ones column: its accumulator (N1) is no margin
        {
            float sqd[NSL], er[NSL];
#pragma unroll
            for (int i = 0; i < NSL; i += 2) {
                const float4 v = *reinterpret_cast<const float4*>(trm + kb * NSL + i);
                sqd[i] = v.x;
                er[i] = v.y;
                sqd[i + 1] = v.z;
                er[i + 1] = v.w;
            }
            uint64_t hit[NSL];
            if (!(KGWAS_MXS_ABLATE & 1)) {
                float mx[NSL];
#pragma unroll
                for (int i = 0; i < NSL; i++) {
                    const float v = acc[i >> 2][CT - 1][i & 3];
                    mx[i] = ones_lane ? 0.0f : fabsf(v);
                }
#pragma unroll
                for (int t = 0; t + 1 < CT; t++)
#pragma unroll
                    for (int i = 0; i < NSL; i++) mx[i] = fmaxf(mx[i], fabsf(acc[i >> 2][t][i & 3]));
#pragma unroll
                for (int i = 0; i < NSL; i++) hit[i] = __ballot(fmaf(-al_min, sqd[i], mx[i]) + er[i] >= 0.0f);
            } else {
#pragma unroll
                for (int i = 0; i < NSL; i++) hit[i] = 0;
                int x = 0;  // keeps every accumulator (and its MFMAs) alive at one lane-op each
#pragma unroll
                for (int i = 0; i < NSL; i++)
#pragma unroll
                    for (int t = 0; t < CT; t++) x ^= __float_as_int(acc[i >> 2][t][i & 3]);
                hit[0] = __ballot(x == 0x7fffffff);
            }
            uint64_t hit_any = 0;
#pragma unroll
            for (int i = 0; i < NSL; i++) hit_any |= hit[i];
            if (hit_any) {  // wave-uniform
                uint32_t mb[CT];  // mb[t] bit i = pair (row slot i, column t*16 + m) survives
#pragma unroll
                for (int t = 0; t < CT; t++) mb[t] = 0;
#pragma unroll
                for (int i = 0; i < NSL; i++) {
                    if (hit[i]) {  // wave-uniform
#pragma unroll
                        for (int t = 0; t < CT; t++) {
                            float al = alc[t];
                            asm volatile("" : "+v"(al));
                            mb[t] |= (fmaf(-al, sqd[i], fabsf(acc[i >> 2][t][i & 3])) + er[i] >= 0.0f) ? (1u << i) : 0u;
                        }
                    }
                }
                // Column (t, m)'s row bits: bit i = 4 rt + jj is row rt * 16 + 4 kb + jj of the wave's rows. In the bitmap's
                // nibble-transposed 64-row words (quarter kb, nibble rt' = row / 16) a wave with four row tiles stores its 16
                // bits as quarter kb, one with two row tiles its 8 bits as byte (rbase / 32) % 2 of that quarter.
                if (RT == 4) {
                    unsigned short* bm16 = reinterpret_cast<unsigned short*>(a.bitmap) + (rbase >> 6) * 4u + kb;
#pragma unroll
                    for (int t = 0; t < CT; t++)
                        if (mb[t]) bm16[(uint64_t)colp[t * 16 + m] * a.words_per_col * 4u] = (unsigned short)mb[t];  // column >= 0 wherever a bit is set
                } else {
                    unsigned char* bm8 = reinterpret_cast<unsigned char*>(a.bitmap) + (rbase >> 6) * 8u + kb * 2u + ((rbase >> 5) & 1u);
#pragma unroll
                    for (int t = 0; t < CT; t++)
                        if (mb[t]) bm8[(uint64_t)colp[t * 16 + m] * a.words_per_col * 8u] = (unsigned char)mb[t];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the exchange area is rewritten by the next pass
#if KGWAS_MXS_PROF
        pf_epi += __builtin_readcyclecounter() - te0;
#endif
    }
    if (!late)
        for (uint32_t i = 0; i < n_dummy; i++) dummy_step();
#if KGWAS_MXS_PROF
    if (threadIdx.x == 0) {
        atomicAdd(&mxs_prof[0], pf_sync);
        atomicAdd(&mxs_prof[1], pf_step);
        atomicAdd(&mxs_prof[2], pf_epi);
        atomicAdd(&mxs_prof[3], pf_nstep);
        atomicAdd(&mxs_prof[4], (unsigned long long)n_passes);
        atomicAdd(&mxs_prof[5], __builtin_readcyclecounter() - pf_k0);
        atomicAdd(&mxs_prof[6], pf_bar);
        atomicAdd(&mxs_prof[7], pf_issue);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stream's last transfers (never read) must not outlive the block's LDS
    if (a.tested) {
        uint32_t v = tested_local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0u && v) atomicAdd(&a.tested[blockIdx.x % TESTED_SHARDS], (unsigned long long)v);
    }
}

// ring + the per-column constants + the waves' row-term exchange areas
// ring + per-column constants + the waves' row-term exchange areas + the fillers' scratch + the row staging areas
static size_t mxs_lds_bytes_t(uint32_t cta, uint32_t rt, uint32_t ng, uint32_t th, uint32_t ring) {
    const uint32_t waves = th / 64u;
    return (size_t)ring * ((cta * MXS_SB + 1023u) / 1024u * 1024u) + 3u * cta * 16u * 4u + waves * (rt * 192u) + waves * 256u + (KGWAS_MXS_DEPHASE ? waves : waves / ng) * (rt * 1024u);
}

template <int CT, int RT, int NG, int TH>
static hipError_t launch_mxs_t(const MxArgs& a, uint32_t rows_per_block, hipStream_t st) {
    const uint32_t rpp = (TH / 64 / NG) * RT * 16u;
    rows_per_block = (rows_per_block + rpp - 1) / rpp * rpp;
    const uint32_t n_rowblocks = (uint32_t)((a.n_rows + rows_per_block - 1) / rows_per_block);
    // four ring slots (slabs and row requests three steps ahead) where they fit the 160 KB, else three
    const bool ring4 = mxs_lds_bytes_t(CT * NG, RT, NG, TH, 4) <= 160u * 1024u;
    const size_t lds = mxs_lds_bytes_t(CT * NG, RT, NG, TH, ring4 ? 4 : 3);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    const void* fn = ring4 ? (const void*)mxs_kernel<CT, RT, NG, TH, 4> : (const void*)mxs_kernel<CT, RT, NG, TH, 3>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t grid = a.n_lgroups > 1 ? (n_rowblocks + 7u) / 8u * 8u * a.n_lgroups : n_rowblocks;
    if (ring4)
        launch_last(mxs_kernel<CT, RT, NG, TH, 4>, dim3(grid), dim3(TH), lds, st, a, rows_per_block, n_rowblocks);
    else
        launch_last(mxs_kernel<CT, RT, NG, TH, 3>, dim3(grid), dim3(TH), lds, st, a, rows_per_block, n_rowblocks);
    return hipGetLastError();
}

// The streaming form exists for two slices, FP6 + FP4 (the default operand set). Shapes: CT column tiles per column group,
// NG column groups per block (a.Bq: [n_lgroups][step][NG][CT][2560 B]; a.cols: [n_lgroups][NG][CT * 16]):
//   NG = 1, CT = 3..7    eight waves, each 64 rows x CT tiles
//   NG = 2, CT = 4..7    two groups of four waves; the waves w and w + 4 work on the same 64 rows, CT tiles each (8..14 in all)
//   NG = 1, CT = 8..13   form 1: eight waves, each 32 rows x CT tiles; form 2: four waves (one per SIMD), each 64 rows x CT tiles
bool mxs_supported(uint32_t CT, uint32_t NG, uint32_t n_slices, uint32_t s1_fp6) {
    return n_slices == 2 && !s1_fp6 && ((NG == 1 && CT >= 3 && CT <= 13) || (NG == 2 && CT >= 4 && CT <= 7));
}
size_t mxs_lds_bytes(uint32_t CT, uint32_t NG) {
    const uint32_t rt = (NG == 1 && CT > 7) ? 2u : 4u;
    const size_t b4 = mxs_lds_bytes_t(CT * NG, rt, NG, 512, 4);
    return b4 <= 160u * 1024u ? b4 : mxs_lds_bytes_t(CT * NG, rt, NG, 512, 3);
}

hipError_t launch_mxs(const MxArgs& a, uint32_t CT, uint32_t NG, uint32_t form, uint32_t rows_per_block, hipStream_t st) {
    if (a.n_rows == 0) return hipSuccess;
    if (!mxs_supported(CT, NG, a.n_slices, a.s1_fp6)) return hipErrorInvalidValue;
    if ((a.n_rows * a.src.stride_dw + a.src.off_dw + a.src.avail_dw) * 4ull >= (1ull << 32)) return hipErrorInvalidValue;  // 32-bit byte offsets
#ifdef KGWAS_MXS_BENCH_ONLY  // experiments: the shapes of the 2048 x 201 and 1135 x 101 benches only (fast compiles)
    if (NG == 2 && CT == 7) return launch_mxs_t<7, 4, 2, 512>(a, rows_per_block, st);
    if (NG == 1 && CT == 13) return form == 2 ? launch_mxs_t<13, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<13, 2, 1, 512>(a, rows_per_block, st);
    if (NG == 1 && CT == 7) return launch_mxs_t<7, 4, 1, 512>(a, rows_per_block, st);
    return hipErrorInvalidValue;
#else
    if (NG == 2) switch (CT) {
            case 4: return launch_mxs_t<4, 4, 2, 512>(a, rows_per_block, st);
            case 5: return launch_mxs_t<5, 4, 2, 512>(a, rows_per_block, st);
            case 6: return launch_mxs_t<6, 4, 2, 512>(a, rows_per_block, st);
            case 7: return launch_mxs_t<7, 4, 2, 512>(a, rows_per_block, st);
        }
    else switch (CT) {
            case 3: return launch_mxs_t<3, 4, 1, 512>(a, rows_per_block, st);
            case 4: return launch_mxs_t<4, 4, 1, 512>(a, rows_per_block, st);
            case 5: return launch_mxs_t<5, 4, 1, 512>(a, rows_per_block, st);
            case 6: return launch_mxs_t<6, 4, 1, 512>(a, rows_per_block, st);
            case 7: return launch_mxs_t<7, 4, 1, 512>(a, rows_per_block, st);
            case 8: return form == 2 ? launch_mxs_t<8, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<8, 2, 1, 512>(a, rows_per_block, st);
            case 9: return form == 2 ? launch_mxs_t<9, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<9, 2, 1, 512>(a, rows_per_block, st);
            case 10: return form == 2 ? launch_mxs_t<10, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<10, 2, 1, 512>(a, rows_per_block, st);
            case 11: return form == 2 ? launch_mxs_t<11, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<11, 2, 1, 512>(a, rows_per_block, st);
            case 12: return form == 2 ? launch_mxs_t<12, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<12, 2, 1, 512>(a, rows_per_block, st);
            case 13: return form == 2 ? launch_mxs_t<13, 4, 1, 256>(a, rows_per_block, st) : launch_mxs_t<13, 2, 1, 512>(a, rows_per_block, st);
        }
    return hipErrorInvalidValue;
#endif
}

}  // namespace kgwas
