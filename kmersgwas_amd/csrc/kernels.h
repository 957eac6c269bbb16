// kernels.h — host-visible launch interface of the gfx950 kernels (kernels.hip).
// Internal to libkgwas; the public boundary is include/kgwas.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env.h"
#include "launch.h"

namespace kgwas {

// One candidate record shipped device -> host (sparse mode).
struct Cand {
    uint64_t kmer;
    double score;
    uint64_t row;  // global file row index
};

// Where the scorer reads a row's squeezed bits from.
//   direct mode  : the .table rows in place  (stride 2*(1+W_f) dwords, bits start at dword 2)
//   squeezed mode: the squeeze kernel's output (stride 2*W_m dwords, bits start at dword 0)
struct RowSrc {
    const uint32_t* base;
    uint64_t stride_dw;
    uint32_t off_dw;
    uint32_t avail_dw;  // dwords of bit data present in a row
};

// The MAC-passing rows of a launch are counted in TESTED_SHARDS counters (block b adds to counter b % TESTED_SHARDS, the
// host sums them): a single counter takes one device-scope atomic per block or wave, and those serialise at the memory
// side (1.3e5 of them made an 8 M-row launch of the bit-sliced filter ten times slower than its arithmetic).
constexpr uint32_t TESTED_SHARDS = 256;

struct ScoreArgs {
    RowSrc src;
    const uint32_t* dmask;      // [2*W_m] AND mask per squeezed dword (0 where no data exists)
    const uint64_t* file_rows;  // the .table rows (for k-mer ids)
    uint64_t file_stride_w;     // 1 + W_f
    uint64_t n_rows;            // rows in this launch
    uint64_t first_row;         // global file row of row 0
    uint32_t S;                 // phenotyped accessions
    uint32_t W_m;               // 64-bit words per squeezed row (even)
    uint32_t n_pheno;
    uint32_t min_count;
    const float* Yperm;   // VALU kernel: [n_pheno][L] permuted, padded (L = 64*W_m)
    const float* Ymfma;   // MFMA kernel: [n_ctiles][L][16] chain-step major
    const float* sums;    // [n_pheno] float32 sequential sums
    const double* thr;    // [n_pheno] stale heap minima (sparse mode)
    // dense outputs (null in sparse mode)
    double* dense;        // [n_pheno][n_rows]
    uint32_t* n1_out;     // [n_rows] masked popcount
    uint64_t* kmer_out;   // [n_rows]
    // sparse outputs (null in dense mode)
    Cand* cand;           // [n_pheno][cap]
    uint32_t* cand_cnt;   // [n_pheno]
    uint32_t cap;
    unsigned long long* tested;  // MAC-passing rows, accumulated ([TESTED_SHARDS]; the exact scorers use [0])
    // Device-side threshold tracking (sparse mode, optional): every shipped candidate is counted in a
    // per-column histogram over the top bits of its score; thr_update_kernel raises thr[p] between
    // chunks to the largest bin boundary with >= topn[p] candidates at or above it.
    uint32_t* hist;             // [n_pheno][hist_bins] or null
    const uint32_t* hist_base;  // [n_pheno] (bits(score) >> HIST_SHIFT) of bin 0
    uint32_t hist_bins;
    // Output of the coarse path (rescore + compaction): the chunk's candidates in (column, row) order.
    double* so_score;    // [key_cap] or null
    uint64_t* so_kmer;   // [key_cap]
    uint32_t* so_row;    // [key_cap] chunk-local row
};

constexpr int HIST_SHIFT = 44;          // 8 mantissa bits per bin: 0.4 % score resolution
constexpr uint32_t HIST_BINS = 16384;   // 64 binades above the threshold at sparse start

// thr[p] = max(thr[p], thr_host[p], largest bin boundary with >= topn[p] counted scores at or above it).
hipError_t launch_copy_to_host(const void* src, void* dst_dev, size_t bytes, hipStream_t st);  // dst_dev: device address of mapped pinned memory
// device pointers of the (mapped) pinned destination
// a chunk's counts, tested-row shards and current thresholds into mapped host buffers, one launch (n_* = 0: skip that part)
hipError_t launch_chunk_tail(const uint32_t* meta, uint32_t n_meta, uint32_t* h_meta, const unsigned long long* tested, uint32_t n_tested,
                             unsigned long long* h_tested, const double* thr, uint32_t n_thr, double* h_thr, hipStream_t st);
// launch_thr_update and launch_chunk_tail in one launch (block p: column p's threshold, also into h_thr[p] unless h_thr is null;
// block 0: the counts and the tested-row shards)
hipError_t launch_thr_tail(const uint32_t* hist, const uint32_t* hist_base, uint32_t bins, const uint64_t* topn, const double* thr_host, double* thr,
                           uint32_t n_pheno, const uint32_t* meta, uint32_t n_meta, uint32_t* h_meta, const unsigned long long* tested,
                           uint32_t n_tested, unsigned long long* h_tested, double* h_thr, hipStream_t st);
hipError_t launch_records_to_host(const double* sc, const uint64_t* km, const uint32_t* rw, uint32_t n, double* h_sc, uint64_t* h_km, uint32_t* h_rw,
                                  hipStream_t st);
hipError_t launch_thr_update(const uint32_t* hist, const uint32_t* hist_base, uint32_t bins, const uint64_t* topn,
                             const double* thr_host, double* thr, uint32_t n_pheno, hipStream_t st);

// Dense start: thr_a[p] = thr_b[p] = thr_host_copy[p] = the topn[p]-th largest score of column p among the chunk's
// MAC-passing rows (dense: [n_pheno][n_rows]); info[0] = MAC-passing rows, info[1] = 1 if one of their scores is NaN.
// Columns with fewer than topn[p] such rows are left alone.
hipError_t launch_dense_select(const double* dense, const uint32_t* n1, uint32_t n_rows, uint32_t n_pheno, uint32_t S, uint32_t min_count,
                               const uint64_t* topn, double* thr_a, double* thr_b, double* thr_host_copy, uint32_t* info, hipStream_t st);

// Scoring kernels. rows_per_block only matters for the MFMA kernel (multiple of 128).
hipError_t launch_score_valu(const ScoreArgs& a, hipStream_t st);
// nb_full = leading 128-sample blocks whose four dwords all exist in the row and need no masking.
hipError_t launch_score_mfma(const ScoreArgs& a, uint32_t rows_per_block, uint32_t nb_full, hipStream_t st);
size_t mfma_lds_bytes(uint32_t W_m);

// Coarse int8 filter (score_coarse.hip). Survivors are bits of a [column][row] bitmap; launch_bitmap_keys lists them,
// and launch_rescore scores them exactly, column by column in row order (it takes the sparse-mode ScoreArgs, with
// Yperm set).
// Per-slot constants of the coarse filter (score_coarse.hip); a slot is one of the PG*16 operand columns of an LDS
// group. y_i ~ c + u*(254*q0_i + q1_i) (two slices) or c + u*q0_i (one slice), c = sum/N. A pair survives iff
// |Dc| >= sqrt(thr)*kalpha*sqrt(d) - iu*E(N1), constants already rounded in the conservative direction by the host.
struct CoarseCol {
    double kalpha;  // (1 - 2^-19) / (N * u)
    float iu;       // 1 / u, rounded up
    int32_t pheno;  // phenotype column in this slot; -1: padding or the group's ones column (slot PG*16-1 -> N1)
};
struct CoarseArgs {
    RowSrc src;
    uint64_t n_rows;
    uint32_t S, n_pheno, min_count;
    uint32_t n_kgroups;     // 512-sample groups = ceil(W_m / 8)
    uint32_t n_lgroups;     // LDS groups of T/NS x 16 operand columns
    uint32_t n_slices;      // int8 slices per column: 1 or 2
    const int8_t* Bq;       // [n_lgroups][n_kgroups][8][T][64 lanes][16] int8 slices, see score_coarse.hip
    const CoarseCol* cols;  // [n_lgroups][T/NS*16]
    const double* thr;      // [n_pheno]
    unsigned long long* bitmap;  // survivors: [n_pheno][words_per_col] 64-bit words, bit r of word w = chunk row 64 w + r; zeroed by the caller
    uint64_t words_per_col;
    unsigned long long* tested;
    // E(N1) = eg_max + min(rall_max, N1 * rmax_max) bounds |yigi_ref - yc| / u_p for every column (units of Dc,
    // each the maximum over the columns, rounded up): float32 summation error of the reference chains, the larger
    // one-sign sum of the quantisation residuals, the largest residual.
    float eg_max, rall_max, rmax_max;
};
size_t coarse_lds_bytes(uint32_t n_kgroups, uint32_t T);
hipError_t launch_coarse(const CoarseArgs& a, uint32_t T, uint32_t rows_per_block, hipStream_t st);
// Block-scaled coarse filter (score_mx.hip): FP4 table bits x FP6 / FP4 phenotype slices on
// v_mfma_scale_f32_16x16x128_f8f6f4, the slices of a column accumulated into ONE float32 accumulator through the block
// scales; same survivors' bitmap, same per-slot constants (CoarseCol, in accumulator units) and error terms as the int8
// filter. Samples are taken in n_full = S / 512 whole groups of 512 (four MFMA steps each) and n_quarter <= 4 quarter groups of
// 128 (one step each).
struct MxArgs {
    RowSrc src;
    uint64_t n_rows;
    uint32_t S, n_pheno, min_count;
    uint32_t n_full, n_quarter;
    uint32_t n_lgroups;     // LDS groups of CT x 16 operand columns
    uint32_t n_slices;      // 1: FP6 only, 2: FP6 + second slice
    uint32_t s1_fp6;        // second slice in FP6 (else FP4)
    uint32_t scale0;        // E8M0 block scale of the first slice, replicated in all four bytes (the second slice's is 2^0)
    const uint8_t* Bq;      // [n_lgroups][4 n_full + n_quarter steps][CT][step bytes], see score_mx.hip
    const CoarseCol* cols;  // [n_lgroups][CT*16]
    const double* thr;      // [n_pheno]
    unsigned long long* bitmap;  // as CoarseArgs
    uint64_t words_per_col;
    unsigned long long* tested;
    float eg_max, rall_max, rmax_max;  // as CoarseArgs, in accumulator units
};
size_t mx_lds_bytes(uint32_t n_steps, uint32_t CT, uint32_t n_slices, uint32_t s1_fp6);
uint32_t mx_step_bytes_rt(uint32_t n_slices, uint32_t s1_fp6);
uint32_t mx_row_tiles(uint32_t CT);  // 16-row tiles per wave pass for this many column tiles
hipError_t launch_mx(const MxArgs& a, uint32_t CT, uint32_t rows_per_block, hipStream_t st);
// The same filter with its operands STREAMED through an LDS ring instead of resident (score_mxs.hip): the waves of a block keep
// the accumulators of ALL column tiles of an operand group (NG column groups of CT tiles: up to 14 tiles, 222 columns), so a row
// is loaded and expanded once per operand group however many samples there are. a.Bq: [n_lgroups][step][NG][CT][2560 B],
// a.cols: [n_lgroups][NG][CT * 16] (each column group with its own ones column in its last slot); two slices, FP6 + FP4 only.
// form (NG = 1, CT > 7 only): 1 = eight waves of 32 rows, 2 = four waves of 64 rows.
bool mxs_supported(uint32_t CT, uint32_t NG, uint32_t n_slices, uint32_t s1_fp6);
size_t mxs_lds_bytes(uint32_t CT, uint32_t NG);
hipError_t launch_mxs(const MxArgs& a, uint32_t CT, uint32_t NG, uint32_t form, uint32_t rows_per_block, hipStream_t st);
// Survivors (sorted keys, per-column ranges) -> exact candidates compacted in (column, row) order into a.so_score /
// a.so_kmer / a.so_row (HBM); meta[0..P) = candidates per column, meta[P..2P) = their offsets, meta[2P] = total,
// meta[2P + 1] = survivor keys emitted. tile_pref [P + 1], tile_cnt [key_cap / 256 + P + 1], tmp_score [key_cap].
hipError_t launch_rescore(const ScoreArgs& a, const uint32_t* keys, const uint32_t* surv_off, const uint32_t* surv_cnt,
                          uint32_t row_bits, uint32_t* tile_pref, uint32_t* tile_cnt, double* tmp_score,
                          const uint32_t* key_count, uint32_t* meta, hipStream_t st);
// Scans of a few columns: every survivor's record goes straight to its place in the key order (a.so_score = exact score or
// -inf for a survivor that is no candidate - the host's replay skips those), no tile scan, no compaction; meta was written
// by launch_narrow_keys (records per column = survivors per column).
hipError_t launch_rescore_direct(const ScoreArgs& a, const uint32_t* keys, const uint32_t* surv_off, const uint32_t* surv_cnt,
                                 uint32_t row_bits, const uint32_t* tile_pref, hipStream_t st);
// Counters of the next chunk and its survivors' bitmap (bitmap_words 64-bit words, may be 0) zeroed in one launch;
// seg_cnt (n_seg_words, may be 0): the narrow filter's per-segment survivor counts. tu.hist != null: the kernel's first
// blocks also raise the thresholds from the histograms (thr_update_kernel's work without a launch of its own: the chunk
// that follows is filtered against everything counted before it).
struct PrepThr {
    const uint32_t* hist;
    const uint32_t* hist_base;
    uint32_t bins;
    const uint64_t* topn;
    const double* thr_host;
    double* thr;
};
hipError_t launch_chunk_prep(uint32_t* cand_cnt, uint32_t n_pheno, unsigned long long* tested, uint32_t* key_count,
                             unsigned long long* bitmap, uint64_t bitmap_words, uint32_t* seg_cnt, uint32_t n_seg_words, const PrepThr& tu,
                             hipStream_t st);

// Narrow filter (score_narrow.hip): scans with one to four phenotype columns. FP4 table bits x three FP8 slices per
// column on v_mfma_scale_f32_16x16x128_f8f6f4; operand row 4 p + k = slice k of phenotype column p, row 4 p + 3 = ones
// (N1), so that lane (table row, p) of the accumulator tile holds everything its pair needs. Survivors go to the same
// key list as the coarse filter's.
constexpr int NARROW_SLICES = 3;
constexpr uint32_t NARROW_MAX_COLS = 4;
struct NarrowCol {
    double w[NARROW_SLICES];  // 2 * u_k: the accumulators are in units of 0.5
    double t1;                // N * c - sum  (c = sum / N rounded to double: tiny)
    double eg, rall, rmax;    // E(N1) = eg + min(rall, N1 * rmax) >= |yigi_ref - yc|, rounded up (phenotype units)
    double pad;               // absolute slack for the double-precision evaluation on both sides
    float wf[3];              // float32 pre-screen: the slice weights ...
    float slackf;             // ... and everything the pre-screen leaves out, as an upper bound on |r| (rounded up)
};
struct NarrowArgs {
    RowSrc src;
    uint64_t n_rows;
    uint32_t S, n_pheno, min_count;
    uint32_t n_kgroups;        // 512-sample groups
    const uint8_t* Bn;         // [n_kgroups * 4 steps][64 lanes][32] FP8 E4M3 slice operands, see score_narrow.hip
    const NarrowCol* cols;     // [n_pheno]
    const double* thr;         // [n_pheno]
    unsigned long long* bitmap;  // survivors: [n_pheno][words_per_col] 64-bit words, bit r of word w = chunk row 64 w + r; zeroed by the caller
    uint64_t words_per_col;
    unsigned long long* tested;
    uint32_t row_off;          // chunk row of this launch's first row (a launch that starts inside a chunk), a multiple of 64
    uint32_t* seg_cnt;         // [n_pheno][n_segs] survivors per 65 536-row segment of the chunk (atomic adds; zeroed by the caller), or null
    uint32_t n_segs;
    uint64_t slack_rows;       // rows of the same buffer that follow the launch's rows (readable: the staged kernel may read whole KB past its last row)
    uint32_t pack1;            // one column whose operand rows are replicated in all four column slots: lane (r, kb) tests row 16 kb + r of the pass
};
// The bitmap's set bits as row-ordered keys (column << row_bits | row), column after column, with each column's range
// (surv_off, surv_cnt) and the total (key_count; above key_cap = overflow, the ranges are then emptied). No sort: counts
// per 65 536-row block, a scan, a scatter. nibble_transposed: the words come from the int8 filters (quarter word kg,
// nibble rt = rows 16 rt + 4 kg ..+3). tile_pref[n_pheno + 1]: launch_rescore's tile table (first tile of each column). blk_scratch: n_pheno * (ceil(n_rows / 65536) + 1) words;
// blk_mask: 4 x 64 bits per (column, block) - the count kernel marks the threads whose words hold a survivor, the scatter kernel
// loads only those and CLEARS them: the bitmap is all zero again behind this call (the caller zeroes it once, not per chunk).
hipError_t launch_bitmap_keys(unsigned long long* bitmap, uint64_t words_per_col, uint64_t n_rows, uint32_t n_pheno, uint32_t* blk_scratch,
                              unsigned long long* blk_mask, uint32_t* keys_sorted, uint32_t key_cap, uint32_t row_bits, uint32_t* surv_off,
                              uint32_t* surv_cnt, uint32_t* key_count, uint32_t* tile_pref, bool nibble_transposed, hipStream_t st);
// The same for the narrow filter (one to four columns), whose kernels have already counted the survivors per segment
// (NarrowArgs::seg_cnt): ONE launch instead of four - every (segment, column) block adds up the counts before it and
// writes its keys; block (0, 0) writes the ranges, the tile table and meta (records = survivors: see launch_rescore_direct).
hipError_t launch_narrow_keys(const unsigned long long* bitmap, uint64_t words_per_col, uint64_t n_rows, uint32_t n_pheno, const uint32_t* seg_cnt,
                              uint32_t* keys_sorted, uint32_t key_cap, uint32_t row_bits, uint32_t* surv_off, uint32_t* surv_cnt,
                              uint32_t* key_count, uint32_t* tile_pref, uint32_t* meta, const unsigned long long* tested_shards, hipStream_t st);
size_t narrow_lds_bytes(uint32_t n_kgroups);
hipError_t launch_narrow(const NarrowArgs& a, uint32_t rows_per_block, hipStream_t st);

// --pattern_counter: append hash_presence_absence_pattern of every MAC-passing row to out[*out_count ...];
// count_distinct_u64 sorts the collected hashes in place (device) and returns how many are distinct.
hipError_t launch_pattern_hash(const RowSrc& src, const uint32_t* dmask, uint64_t n_rows, uint32_t S, uint32_t W_m,
                               uint32_t min_count, uint64_t* out, unsigned long long* out_count, hipStream_t st);
hipError_t count_distinct_u64(uint64_t* keys, uint64_t n, uint64_t* result, hipStream_t st);

// kmers_table_to_bed (bed_kernels.hip): per squeezed row its popcount and pattern hash; the PLINK bytes.
hipError_t launch_bed_rowinfo(const uint32_t* sq, uint64_t n_rows, uint32_t W_m, uint32_t* n1_out, uint64_t* hash_out,
                              hipStream_t st);
hipError_t launch_bed_bytes(const uint32_t* sq, uint64_t n_rows, uint32_t W_m, uint32_t bytes_per_row, uint8_t* out,
                            hipStream_t st);

// SNP scorer (snp_kernels.hip): .bed bytes -> three bit planes per SNP ([snp][3][ndw]); planes -> scores[p][snp].
hipError_t launch_snp_planes(const uint8_t* bed, uint64_t n_snps, uint32_t bytes_per_snp, const uint32_t* byte_idx,
                             const uint32_t* shift, uint32_t S, uint32_t ndw, uint32_t* planes, hipStream_t st);
hipError_t launch_snp_score(const uint32_t* planes, uint64_t n_snps, uint32_t ndw, const float* Yperm, uint32_t L, uint32_t n_pheno,
                            double mac, double* scores, hipStream_t st);

// Squeeze: out[r][2*W_m dwords] bit i = file bit colmap[i] (colmap[i] == 0xFFFFFFFF -> 0).
hipError_t launch_squeeze(const uint64_t* file_rows, uint64_t file_stride_w, uint64_t n_rows, const uint32_t* colmap,
                          uint32_t W_m, uint32_t W_f, uint32_t* out, hipStream_t st);

// Synthetic rows.
hipError_t launch_synth(uint64_t* rows, uint64_t first_row, uint64_t n_rows, uint64_t n_acc, uint64_t seed,
                        hipStream_t st);

// Kinship: transpose+filter, then Hamming Gram accumulation into H (S_pad x S_pad u64).
//  T: [S_pad][n_rw] u32, n_rw = ceil(n_rows/64)*2
hipError_t launch_kin_transpose(const uint64_t* file_rows, uint64_t file_stride_w, uint64_t n_rows, uint32_t S_f,
                                uint32_t S_pad, uint32_t min_count, uint32_t* T, uint64_t n_rw,
                                unsigned long long* n_used, hipStream_t st);
// rows per transpose block the LDS allows for this table shape (256, 128, 64; 0 = too many accessions)
uint32_t kin_transpose_rows_per_block(uint64_t file_stride_w, uint32_t S_pad);
// scratch: kin_gram_scratch_bytes(S_pad) bytes (the row slices' partial tiles, reduced into H by a second kernel)
size_t kin_gram_scratch_bytes(uint32_t S_pad);
hipError_t launch_kin_gram(const uint32_t* T, uint64_t n_rw, uint32_t S_pad, unsigned long long* H, void* scratch, size_t scratch_bytes,
                           hipStream_t st);

}  // namespace kgwas
