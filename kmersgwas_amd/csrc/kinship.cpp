// kinship.cpp — emma_kinship_kmers' accumulation (src/emma_kinship_kmers.cpp:77-102,
// src/kmers_multiple_databases.cpp:418-438) as two GPU kernels per chunk:
//   kin_transpose: MAC filter over all S_f columns + bit transpose (sample-major bit planes)
//   kin_gram     : Hamming distances H[i][j] += popcount(T_i ^ T_j)
// and the closed form K[i][j] = sum_rows (1 ^ g_i ^ g_j) = n_used - H[i][j]. Everything is
// integer, so partial results of different chunks / shards / GPUs simply add.
// Also hosts the synthetic-row entry points.
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"
#include "ingest.h"
#include "kernels.h"
#include "synth.h"

using namespace kgwas;

struct kgwas_kinship {
    int device = 0;
    uint64_t S_f = 0, W_f = 0, min_count = 0;
    uint32_t S_pad = 0;
    uint64_t chunk_rows = 0, n_rw_cap = 0;
    hipStream_t stream = nullptr, stream_tr = nullptr;  // Gram accumulation / bit transposes
    hipEvent_t ev_user = nullptr, ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_tr[2] = {nullptr, nullptr}, ev_gram[2] = {nullptr, nullptr};
    bool gram_pending[2] = {false, false};
    uint32_t* d_T2[2] = {nullptr, nullptr};  // sample-major bit planes of two chunks
    unsigned long long* d_H = nullptr;
    void* d_part = nullptr;  // partial tiles of a chunk's row slices (kin_gram_scratch_bytes)
    size_t part_bytes = 0;
    unsigned long long* d_n = nullptr;
    Ingest ingest;  // host / file feeds: pinned and device piece rings, a copy stream (ingest.h)
    double kernel_ms = 0;
    uint64_t launches = 0, rows_fed = 0;
    ~kgwas_kinship() {
        (void)hipSetDevice(device);
        for (int b = 0; b < 2; b++) {
            if (d_T2[b]) (void)hipFree(d_T2[b]);
            if (ev_tr[b]) (void)hipEventDestroy(ev_tr[b]);
            if (ev_gram[b]) (void)hipEventDestroy(ev_gram[b]);
        }
        if (stream_tr) (void)hipStreamDestroy(stream_tr);
        if (d_part) (void)hipFree(d_part);
        if (d_H) (void)hipFree(d_H);
        if (d_n) (void)hipFree(d_n);
        if (ev_user) (void)hipEventDestroy(ev_user);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Chunks of 2^20 rows: the transpose of chunk i + 1 (memory- and LDS-bound) runs on its own stream beside the Gram
// accumulation of chunk i (matrix-pipe-bound) - two plane buffers, two events per buffer.
static void kin_feed(kgwas_kinship* k, const uint64_t* d_rows, uint64_t n_rows) {
    const uint64_t stride = 1 + k->W_f;
    if (n_rows == 0) return;
    // the caller's rows are ready on k->stream (kgwas_kinship_feed_device ordered it after the caller's stream)
    KGWAS_HIP(hipEventRecord(k->ev0, k->stream));
    KGWAS_HIP(hipStreamWaitEvent(k->stream_tr, k->ev0, 0));
    uint64_t i = 0;
    for (uint64_t pos = 0; pos < n_rows; pos += k->chunk_rows, i++) {
        const uint64_t c = std::min<uint64_t>(k->chunk_rows, n_rows - pos);
        const uint64_t n_rw = (c + 511) / 512 * 16;  // u32 words per sample, whole 512-row blocks
        const int b = (int)(i & 1);
        if (k->gram_pending[b]) KGWAS_HIP(hipStreamWaitEvent(k->stream_tr, k->ev_gram[b], 0));  // buffer b is free again
        static const bool no_tr = exp_set("KGWAS_KIN_NO_TR");  // experiments: the Gram kernel alone (wrong results)
        if (!no_tr || i < 2)
        KGWAS_HIP(launch_kin_transpose(d_rows + pos * stride, stride, c, (uint32_t)k->S_f, k->S_pad,
                                       (uint32_t)std::min<uint64_t>(k->min_count, 0xFFFFFFFFull), k->d_T2[b], n_rw, k->d_n,
                                       k->stream_tr));
        KGWAS_HIP(hipEventRecord(k->ev_tr[b], k->stream_tr));
        KGWAS_HIP(hipStreamWaitEvent(k->stream, k->ev_tr[b], 0));
        KGWAS_HIP(launch_kin_gram(k->d_T2[b], n_rw, k->S_pad, k->d_H, k->d_part, k->part_bytes, k->stream));
        KGWAS_HIP(hipEventRecord(k->ev_gram[b], k->stream));
        k->gram_pending[b] = true;
        k->launches++;
    }
    KGWAS_HIP(hipEventRecord(k->ev1, k->stream));
    KGWAS_HIP(hipStreamSynchronize(k->stream));  // the caller may reuse its rows when a feed returns
    KGWAS_HIP(hipStreamSynchronize(k->stream_tr));
    k->gram_pending[0] = k->gram_pending[1] = false;
    float ms = 0;
    KGWAS_HIP(hipEventElapsedTime(&ms, k->ev0, k->ev1));
    k->kernel_ms += ms;
    k->rows_fed += n_rows;
}

extern "C" {

int kgwas_kinship_create(int32_t device, uint64_t n_acc_file, uint64_t min_count, kgwas_kinship** out) {
    return guarded([&] {
        if (!out || n_acc_file == 0 || n_acc_file >= (1ull << 31)) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_create: bad argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0)
            throw Error(KGWAS_ERR_DEVICE, "no HIP device available: libkgwas has no CPU fallback");
        if (device < 0 || device >= n) throw Error(KGWAS_ERR_ARG, "device ordinal out of range");
        KGWAS_HIP(hipSetDevice(device));
        std::unique_ptr<kgwas_kinship> k(new kgwas_kinship);
        k->device = device;
        k->S_f = n_acc_file;
        k->W_f = (n_acc_file + 63) / 64;
        k->min_count = min_count;
        k->S_pad = (uint32_t)((n_acc_file + 127) / 128 * 128);  // whole 128 x 128 Gram tiles
        if (!kin_transpose_rows_per_block(1 + k->W_f, k->S_pad))
            throw Error(KGWAS_ERR_ARG, "kgwas_kinship_create: " + std::to_string(n_acc_file) +
                                           " accessions exceed what the bit-transpose kernel can stage in LDS (about 9900)");
        k->chunk_rows = 1ull << 20;  // the reference's own batch size (src/emma_kinship_kmers.cpp:89)
        k->n_rw_cap = (k->chunk_rows + 511) / 512 * 16;
        KGWAS_HIP(hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking));
        KGWAS_HIP(hipEventCreate(&k->ev_user));
        KGWAS_HIP(hipEventCreate(&k->ev0));
        KGWAS_HIP(hipEventCreate(&k->ev1));
        KGWAS_HIP(hipStreamCreateWithFlags(&k->stream_tr, hipStreamNonBlocking));
        for (int b = 0; b < 2; b++) {
            KGWAS_HIP(hipMalloc((void**)&k->d_T2[b], (size_t)k->S_pad * k->n_rw_cap * 4));
            KGWAS_HIP(hipEventCreateWithFlags(&k->ev_tr[b], hipEventDisableTiming));
            KGWAS_HIP(hipEventCreateWithFlags(&k->ev_gram[b], hipEventDisableTiming));
        }
        KGWAS_HIP(hipMalloc((void**)&k->d_H, (size_t)k->S_pad * k->S_pad * 8));
        k->part_bytes = kin_gram_scratch_bytes(k->S_pad);
        KGWAS_HIP(hipMalloc(&k->d_part, k->part_bytes));
        KGWAS_HIP(hipMalloc((void**)&k->d_n, TESTED_SHARDS * 8));
        KGWAS_HIP(hipMemset(k->d_H, 0, (size_t)k->S_pad * k->S_pad * 8));
        KGWAS_HIP(hipMemset(k->d_n, 0, TESTED_SHARDS * 8));
        *out = k.release();
    });
}

int kgwas_kinship_feed_device(kgwas_kinship* k, const void* d_rows, uint64_t n_rows, void* hip_stream) {
    return guarded([&] {
        if (!k || (!d_rows && n_rows)) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_feed_device: null argument");
        KGWAS_HIP(hipSetDevice(k->device));
        KGWAS_HIP(hipEventRecord(k->ev_user, (hipStream_t)hip_stream));
        KGWAS_HIP(hipStreamWaitEvent(k->stream, k->ev_user, 0));
        kin_feed(k, reinterpret_cast<const uint64_t*>(d_rows), n_rows);
    });
}

int kgwas_kinship_feed_host(kgwas_kinship* k, const uint64_t* rows, uint64_t n_rows) {
    return guarded([&] {
        if (!k || (!rows && n_rows)) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_feed_host: null argument");
        KGWAS_HIP(hipSetDevice(k->device));
        const uint64_t stride = 1 + k->W_f;
        k->ingest.file_feed_ = false;
        k->ingest.run(
            stride, n_rows, k->chunk_rows, k->stream,
            [&](uint64_t* dst, uint64_t row_off, uint64_t cnt) { memcpy(dst, rows + row_off * stride, cnt * stride * 8); },
            [&](const uint64_t* d_rows, uint64_t, uint64_t cnt) { kin_feed(k, d_rows, cnt); });
    });
}

int kgwas_kinship_feed_table(kgwas_kinship* k, kgwas_table* t, uint64_t row0, uint64_t n_rows) {
    return guarded([&] {
        if (!k || !t) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_feed_table: null argument");
        uint64_t n_acc = 0, t_rows = 0, wpr = 0;
        uint32_t klen = 0;
        if (kgwas_table_info(t, &n_acc, &t_rows, &wpr, &klen) != KGWAS_OK) throw Error(KGWAS_ERR_ARG, kgwas_last_error());
        if (n_acc != k->S_f) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_feed_table: table and session disagree on the accession count");
        if (row0 > t_rows || n_rows > t_rows - row0) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_feed_table: out of range");
        KGWAS_HIP(hipSetDevice(k->device));
        k->ingest.file_feed_ = true;
        k->ingest.run(
            1 + k->W_f, n_rows, k->chunk_rows, k->stream,
            [&](uint64_t* dst, uint64_t row_off, uint64_t cnt) {
                if (kgwas_table_read_rows(t, row0 + row_off, cnt, dst) != KGWAS_OK) throw Error(KGWAS_ERR_IO, kgwas_last_error());
            },
            [&](const uint64_t* d_rows, uint64_t, uint64_t cnt) { kin_feed(k, d_rows, cnt); });
    });
}

int kgwas_kinship_partials(kgwas_kinship* k, uint64_t* hamming, uint64_t* n_used) {
    return guarded([&] {
        if (!k || !hamming || !n_used) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_partials: null argument");
        KGWAS_HIP(hipSetDevice(k->device));
        KGWAS_HIP(hipStreamSynchronize(k->stream));
        // (written into pinned, mapped memory by the GPU: a process's first SDMA device -> host transfer blocks its call for
        // ~60 ms - a fifth of an `emma_kinship_kmers` run on 40 M rows -, and the copy into pageable memory is staged anyway)
        PinBuf<unsigned long long> Hp;
        Hp.alloc((size_t)k->S_pad * k->S_pad);
        KGWAS_HIP(launch_copy_to_host(k->d_H, Hp.dev(), Hp.n * 8, k->stream));
        KGWAS_HIP(hipStreamSynchronize(k->stream));
        const unsigned long long* H = Hp.p;
        unsigned long long n = 0, shards[TESTED_SHARDS];
        KGWAS_HIP(hipMemcpy(shards, k->d_n, sizeof(shards), hipMemcpyDeviceToHost));
        for (unsigned long long v : shards) n += v;
        // d_H holds the Gram counts c_ij = sum_rows g_i g_j (tiles on or above the diagonal of the 128-sample tiling);
        // Hamming(i, j) = c_ii + c_jj - 2 c_ij
        auto c = [&](uint64_t i, uint64_t j) { return (i / 128 <= j / 128) ? H[i * k->S_pad + j] : H[j * k->S_pad + i]; };
        for (uint64_t i = 0; i < k->S_f; i++)
            for (uint64_t j = 0; j < k->S_f; j++) hamming[i * k->S_f + j] = (i == j) ? 0 : c(i, i) + c(j, j) - 2 * c(i, j);
        *n_used = n;
    });
}

int kgwas_kinship_from_partials(uint64_t n_acc, const uint64_t* hamming, uint64_t n_used, uint64_t* K) {
    return guarded([&] {
        if (!hamming || !K) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_from_partials: null argument");
        for (uint64_t i = 0; i < n_acc; i++)
            for (uint64_t j = 0; j < n_acc; j++)
                K[i * n_acc + j] = (j < i) ? (n_used - hamming[i * n_acc + j]) : 0;  // lower triangle, as :430-432
    });
}

int kgwas_kinship_get_stats(const kgwas_kinship* k, double* kernel_ms, uint64_t* launches, uint64_t* rows_fed) {
    return guarded([&] {
        if (!k) throw Error(KGWAS_ERR_ARG, "kgwas_kinship_get_stats: null");
        if (kernel_ms) *kernel_ms = k->kernel_ms;
        if (launches) *launches = k->launches;
        if (rows_fed) *rows_fed = k->rows_fed;
    });
}

void kgwas_kinship_destroy(kgwas_kinship* k) { delete k; }

// ---- synthetic rows ---------------------------------------------------------------------------
int kgwas_synth_rows_device(void* d_rows, uint64_t first_row, uint64_t n_rows, uint64_t n_acc, uint64_t seed,
                            void* hip_stream) {
    return guarded([&] {
        if ((!d_rows && n_rows) || n_acc == 0) throw Error(KGWAS_ERR_ARG, "kgwas_synth_rows_device: bad argument");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(KGWAS_ERR_DEVICE, "no HIP device available");
        KGWAS_HIP(launch_synth(reinterpret_cast<uint64_t*>(d_rows), first_row, n_rows, n_acc, seed, (hipStream_t)hip_stream));
    });
}

int kgwas_synth_rows_host(uint64_t* rows, uint64_t first_row, uint64_t n_rows, uint64_t n_acc, uint64_t seed) {
    return guarded([&] {
        if ((!rows && n_rows) || n_acc == 0) throw Error(KGWAS_ERR_ARG, "kgwas_synth_rows_host: bad argument");
        const uint64_t W = 1 + (n_acc + 63) / 64;
        for (uint64_t r = 0; r < n_rows; r++)
            for (uint32_t w = 0; w < W; w++) rows[r * W + w] = synth_word(seed, first_row + r, w, n_acc);
    });
}

}  // extern "C"
