"""Host-side mirror of the reference's interface for the association-scan path, over the C ABI.

Names follow the reference's classes (src/kmers_multiple_databases.h, src/best_associations_heap.h):

  KmersTable            <- MultipleKmersDataBases' ctor guards + .names            (a-1, a-2)
  Phenotypes            <- load_phenotypes_file                                    (a-10)
  AssociationScan       <- load_kmers + add_kmers_to_heap over all columns (pass 1) (a-3 .. a-8)
  BestAssociationsHeap  <- BestAssociationsHeap                                    (a-7)
  Kinship               <- update_emma_kinshhip_calculation / emma_kinship_kmers    (a-9)

Everything numeric happens inside libkgwas (HIP kernels on the GPU + the std::priority_queue
replay); this module only moves buffers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Sequence

import numpy as np

from . import capi
from .capi import lib, check, ptr


def min_count(n_acc: int, maf: float, mac: int) -> int:
    return int(lib.kgwas_min_count(n_acc, maf, mac))


class KmersTable:
    def __init__(self, base: str, kmer_len: int = 0):
        self._h = C.c_void_p()
        check(lib.kgwas_table_open(base.encode(), kmer_len, C.byref(self._h)))
        a, r, w, k = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(lib.kgwas_table_info(self._h, C.byref(a), C.byref(r), C.byref(w), C.byref(k)))
        self.base = base
        self.n_acc, self.n_rows, self.words_per_row, self.kmer_len = a.value, r.value, w.value, k.value
        self.names = []
        for i in range(self.n_acc):
            s = C.c_char_p()
            check(lib.kgwas_table_name(self._h, i, C.byref(s)))
            self.names.append(s.value.decode())

    def column_map(self, accessions: Sequence[str]) -> np.ndarray:
        arr = (C.c_char_p * len(accessions))(*[a.encode() for a in accessions])
        out = np.zeros(len(accessions), dtype=np.uint64)
        check(lib.kgwas_table_column_map(self._h, arr, len(accessions), out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def read_rows(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, 1 + self.words_per_row), dtype=np.uint64)
        check(lib.kgwas_table_read_rows(self._h, row0, n, ptr(out)))
        return out

    def close(self):
        if self._h:
            lib.kgwas_table_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()


class Phenotypes:
    def __init__(self, path: str):
        h = C.c_void_p()
        check(lib.kgwas_pheno_load(path.encode(), C.byref(h)))
        try:
            p, a = C.c_uint64(), C.c_uint64()
            check(lib.kgwas_pheno_info(h, C.byref(p), C.byref(a)))
            self.names, self.accessions = [], []
            s = C.c_char_p()
            for j in range(p.value):
                check(lib.kgwas_pheno_name(h, j, C.byref(s)))
                self.names.append(s.value.decode())
            for i in range(a.value):
                check(lib.kgwas_pheno_accession(h, i, C.byref(s)))
                self.accessions.append(s.value.decode())
            y = C.POINTER(C.c_float)()
            check(lib.kgwas_pheno_values(h, C.byref(y)))
            n = p.value * a.value
            self.Y = (np.ctypeslib.as_array(y, shape=(n,)).copy() if n else np.zeros(0, np.float32)).reshape(
                p.value, a.value)
        finally:
            lib.kgwas_pheno_free(h)


class BestAssociationsHeap:
    def __init__(self, max_results: int, _handle=None):
        self._h = C.c_void_p(_handle) if _handle else C.c_void_p()
        if not _handle:
            check(lib.kgwas_heap_new(max_results, C.byref(self._h)))

    def add_associations(self, kmer, score, row):
        k = np.ascontiguousarray(kmer, np.uint64)
        s = np.ascontiguousarray(score, np.float64)
        r = np.ascontiguousarray(row, np.uint64)
        check(lib.kgwas_heap_add_many(self._h, ptr(k), ptr(s), ptr(r), len(k)))

    def add_association(self, kmer, score, row):
        self.add_associations([kmer], [score], [row])

    def _size(self):
        n, ins, low = C.c_uint64(), C.c_uint64(), C.c_double()
        check(lib.kgwas_heap_size(self._h, C.byref(n), C.byref(ins), C.byref(low)))
        return n.value, ins.value, low.value

    def __len__(self):
        return self._size()[0]

    @property
    def lowest_score(self):
        return self._size()[2]

    def pop_all(self):
        n = len(self)
        k, s, r = np.zeros(n, np.uint64), np.zeros(n, np.float64), np.zeros(n, np.uint64)
        check(lib.kgwas_heap_pop_all(self._h, ptr(k), ptr(s), ptr(r)))
        return k, s, r

    def get_kmers_for_output(self):
        n = len(self)
        k, rk, r = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        check(lib.kgwas_heap_output_list(self._h, ptr(k), ptr(rk), ptr(r)))
        return k, rk, r

    def get_rows_sorted_indices(self):
        """src/best_associations_heap.cpp:135-147."""
        r = np.zeros(len(self), np.uint64)
        check(lib.kgwas_heap_rows_sorted(self._h, ptr(r)))
        return r

    def output_to_file(self, path: str, with_scores: bool = False):
        """output_to_file / output_to_file_with_scores (src/best_associations_heap.cpp:65-90)."""
        check(lib.kgwas_heap_output_to_file(self._h, path.encode(), 1 if with_scores else 0))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.kgwas_heap_free(self._h)
            self._h = None


class AssociationScan:
    """Pass 1 of associate_kmers for all phenotype columns at once.
    record_history: what a shard other than the first keeps for a cross-shard merge (kmersgwas_amd.dist): 1 / True =
    every effective heap push (history(), history_above()); 2 = each heap's last evictions only (history_above()
    alone, 6 % of a pass instead of 14 %; raises if the ring - KGWAS_HISTORY_RING, default 4096 - was too short)."""

    def __init__(self, n_acc_file: int, col, Y, topn, min_count: int, device: int = 0, kernel: int = capi.KERNEL_AUTO,
                 chunk_rows: int = 0, host_threads: int = 0, record_history: int = 0,
                 count_patterns: bool = False):
        self.col = np.ascontiguousarray(col, np.uint64)
        self.Y = np.ascontiguousarray(Y, np.float32)
        if self.Y.ndim == 1:
            self.Y = self.Y[None, :]
        self.n_pheno, self.n_acc = self.Y.shape
        self.topn = np.ascontiguousarray(np.broadcast_to(np.asarray(topn, np.uint64), (self.n_pheno,)))
        self.n_acc_file = n_acc_file
        self.words_per_row = (n_acc_file + 63) // 64
        p = capi.ScanParams()
        p.struct_size = C.sizeof(capi.ScanParams)
        p.device = device
        p.n_acc_file = n_acc_file
        p.n_acc = self.n_acc
        p.col = self.col.ctypes.data_as(C.POINTER(C.c_uint64))
        p.n_pheno = self.n_pheno
        p.Y = self.Y.ctypes.data_as(C.POINTER(C.c_float))
        p.topn = self.topn.ctypes.data_as(C.POINTER(C.c_uint64))
        p.min_count = min_count
        p.chunk_rows = chunk_rows
        p.host_threads = host_threads
        p.kernel = kernel
        p.record_history = int(record_history)  # False/0 off, True/1 full log, 2 eviction ring
        p.count_patterns = 1 if count_patterns else 0
        self._h = C.c_void_p()
        check(lib.kgwas_scan_create(C.byref(p), C.byref(self._h)))
        # columns may be finished by selection instead of the push-by-push replay (kmersgwas_amd/csrc/scan_lazy.cpp): what
        # kmersgwas_amd.dist.merge_shards looks at (it must come out the same on every rank of a merge). Asked of the library:
        # sessions that fell back to the exact scorers (non-finite phenotypes, an explicit kernel) are NOT in select mode.
        on = C.c_int(0)
        check(lib.kgwas_scan_select_mode(self._h, C.byref(on)))
        self.select_mode = bool(on.value)

    # rows: uint64 array (n_rows, 1 + W_f) in host memory
    def feed_host(self, rows: np.ndarray, first_row: int = 0):
        rows = np.ascontiguousarray(rows, np.uint64)
        n = rows.size // (1 + self.words_per_row)
        check(lib.kgwas_scan_feed_host(self._h, ptr(rows), n, first_row))

    def feed_table(self, table: "KmersTable", row0: int, n_rows: int):
        """Rows [row0, row0 + n_rows) of an open table, read / copied / scored in overlapping 128 MiB pieces."""
        check(lib.kgwas_scan_feed_table(self._h, table._h, row0, n_rows))

    # device pointer to rows already resident in HBM (e.g. torch tensor .data_ptr())
    def expect_finish(self):
        """Hint: the next feed is the last one before finish(); the replay workers that run out of records near its end
        then pop complete columns into the result lists while the slowest worker is still replaying (results unchanged)."""
        check(lib.kgwas_scan_expect_finish(self._h))

    def feed_device(self, d_ptr: int, n_rows: int, first_row: int = 0, stream: int = 0):
        check(lib.kgwas_scan_feed_device(self._h, C.c_void_p(d_ptr), n_rows, first_row, C.c_void_p(stream)))

    def finish(self):
        check(lib.kgwas_scan_finish(self._h))

    def reset(self):
        """Empty heaps / histories / statistics, keep all buffers (session reuse)."""
        check(lib.kgwas_scan_reset(self._h))

    def lowest(self):
        """(lowest_score[n_pheno], full[n_pheno]) of the heaps as they stand."""
        low = np.zeros(self.n_pheno, np.float64)
        full = np.zeros(self.n_pheno, np.uint8)
        check(lib.kgwas_scan_lowest(self._h, ptr(low), ptr(full)))
        return low, full.astype(bool)

    def absorb(self, shard_histories):
        """Replay later shards' (pre-filtered) histories, in shard order, into this scan's heaps.
        shard_histories[g][j] = (kmer, score, row). Call finish() again afterwards."""
        G = len(shard_histories)
        if G == 0:
            return
        P = self.n_pheno
        counts = np.zeros((G, P), np.uint64)
        ks, ss, rs = [], [], []
        for g in range(G):
            for j in range(P):
                counts[g, j] = len(shard_histories[g][j][0])
            ks.append(np.ascontiguousarray(np.concatenate([np.asarray(h[0], np.uint64) for h in shard_histories[g]])))
            ss.append(np.ascontiguousarray(np.concatenate([np.asarray(h[1], np.float64) for h in shard_histories[g]])))
            rs.append(np.ascontiguousarray(np.concatenate([np.asarray(h[2], np.uint64) for h in shard_histories[g]])))
        kp = (C.c_void_p * G)(*[a.ctypes.data for a in ks])
        sp = (C.c_void_p * G)(*[a.ctypes.data for a in ss])
        rp = (C.c_void_p * G)(*[a.ctypes.data for a in rs])
        check(lib.kgwas_scan_absorb(self._h, G, ptr(counts), kp, sp, rp))

    def absorb_flat(self, counts: np.ndarray, kmer: Sequence[np.ndarray], score: Sequence[np.ndarray], row: Sequence[np.ndarray]):
        """absorb() on flat arrays: counts[g][j] entries of column j in shard g, shard g's entries concatenated by
        column in kmer[g] / score[g] / row[g]."""
        counts = np.ascontiguousarray(counts, np.uint64)
        G = counts.shape[0]
        if G == 0:
            return
        ks = [np.ascontiguousarray(a, np.uint64) for a in kmer]
        ss = [np.ascontiguousarray(a, np.float64) for a in score]
        rs = [np.ascontiguousarray(a, np.uint64) for a in row]
        kp = (C.c_void_p * G)(*[a.ctypes.data for a in ks])
        sp = (C.c_void_p * G)(*[a.ctypes.data for a in ss])
        rp = (C.c_void_p * G)(*[a.ctypes.data for a in rs])
        check(lib.kgwas_scan_absorb(self._h, G, ptr(counts), kp, sp, rp))

    def _flat3(self, total, k, s, r):
        """Views of the library's export scratch: valid until the next history_above / heaps_export call."""
        if total == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.float64), np.zeros(0, np.uint64)
        return (np.ctypeslib.as_array(k, (total,)), np.ctypeslib.as_array(s, (total,)), np.ctypeslib.as_array(r, (total,)))

    def history_above(self, thr: np.ndarray):
        """Recorded history entries with score > thr[j] (thr = -inf keeps all), flat by column:
        (counts[P], kmer, score, row)."""
        thr = np.ascontiguousarray(thr, np.float64)
        counts = np.zeros(self.n_pheno, np.uint64)
        k, s, r = C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)(), C.POINTER(C.c_uint64)()
        check(lib.kgwas_scan_history_above(self._h, ptr(thr), ptr(counts), C.byref(k), C.byref(s), C.byref(r)))
        return (counts,) + self._flat3(int(counts.sum()), k, s, r)

    def heaps_export(self, cols):
        """State of the heaps `cols` in heap-array order, flat: (sizes[len(cols)], kmer, score, row)."""
        cols = np.ascontiguousarray(cols, np.uint64)
        sizes = np.zeros(len(cols), np.uint64)
        k, s, r = C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)(), C.POINTER(C.c_uint64)()
        check(lib.kgwas_scan_heaps_export(self._h, len(cols), ptr(cols), ptr(sizes), C.byref(k), C.byref(s), C.byref(r)))
        return (sizes,) + self._flat3(int(sizes.sum()), k, s, r)

    @staticmethod
    def _msg_args(col0, ncols, out):
        col0 = np.ascontiguousarray(col0, np.uint64)
        ncols = np.ascontiguousarray(ncols, np.uint64)
        assert len(col0) == len(ncols)
        words = np.zeros(len(col0), np.uint64)
        if out is None:
            return col0, ncols, words, None, 0
        assert out.dtype.itemsize == 8 and out.ndim == 1 and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]
        return col0, ncols, words, C.c_void_p(out.ctypes.data), len(out)

    def history_above_msgs(self, thr: np.ndarray, col0, ncols, out):
        """history_above() as messages written into `out` (a 1-D array of 64-bit words, e.g. a view of a pinned staging
        buffer; include/kgwas.h describes the layout): message m = the columns col0[m] .. col0[m] + ncols[m] - 1.
        Returns the messages' lengths in words; nothing was written if their sum exceeds len(out)."""
        thr = np.ascontiguousarray(thr, np.float64)
        assert len(thr) == self.n_pheno
        col0, ncols, words, p_out, cap = self._msg_args(col0, ncols, out)
        check(lib.kgwas_scan_history_above_msgs(self._h, ptr(thr), len(col0), ptr(col0), ptr(ncols), p_out, cap, ptr(words)))
        return words

    def heaps_export_msgs(self, col0, ncols, out):
        """heaps_export() as messages written into `out` (see history_above_msgs)."""
        col0, ncols, words, p_out, cap = self._msg_args(col0, ncols, out)
        check(lib.kgwas_scan_heaps_export_msgs(self._h, len(col0), ptr(col0), ptr(ncols), p_out, cap, ptr(words)))
        return words

    def heaps_import(self, cols, sizes, kmer, score, row):
        """Re-create exported heap states (layout included) in this session. Call finish() again afterwards."""
        cols = np.ascontiguousarray(cols, np.uint64)
        sizes = np.ascontiguousarray(sizes, np.uint64)
        kmer = np.ascontiguousarray(kmer, np.uint64)
        score = np.ascontiguousarray(score, np.float64)
        row = np.ascontiguousarray(row, np.uint64)
        assert len(kmer) == len(score) == len(row) == int(sizes.sum())
        check(lib.kgwas_scan_heaps_import(self._h, len(cols), ptr(cols), ptr(sizes), ptr(kmer), ptr(score), ptr(row)))

    def _lists(self, fn, j):
        n = C.c_uint64()
        k, s, r = C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)(), C.POINTER(C.c_uint64)()
        check(fn(self._h, j, C.byref(n), C.byref(k), C.byref(s), C.byref(r)))
        m = n.value
        if m == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.float64), np.zeros(0, np.uint64)
        return (np.ctypeslib.as_array(k, shape=(m,)).copy(), np.ctypeslib.as_array(s, shape=(m,)).copy(),
                np.ctypeslib.as_array(r, shape=(m,)).copy())

    def result(self, j: int):
        """(kmers, scores, file rows) of column j in heap-pop order (ascending score; rank of entry i = n - i)."""
        return self._lists(lib.kgwas_scan_result, j)

    def history(self, j: int):
        return self._lists(lib.kgwas_scan_history, j)

    def stats(self) -> dict:
        st = capi.ScanStats()
        check(lib.kgwas_scan_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def scores_dense(self, rows=None, d_ptr: int = 0, n_rows: int = 0):
        """calculate_kmer_score for every row x column. Returns (scores[n_pheno, n_rows], popcnt[n_rows])."""
        if rows is not None:
            rows = np.ascontiguousarray(rows, np.uint64)
            n_rows = rows.size // (1 + self.words_per_row)
            src, on_dev = ptr(rows), 0
        else:
            src, on_dev = C.c_void_p(d_ptr), 1
        sc = np.zeros((self.n_pheno, n_rows), np.float64)
        pc = np.zeros(n_rows, np.uint32)
        check(lib.kgwas_scan_scores_dense(self._h, src, on_dev, n_rows, ptr(sc), ptr(pc)))
        return sc, pc

    def close(self):
        if getattr(self, "_h", None):
            lib.kgwas_scan_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class MultiDeviceScan:
    """kgwas_multiscan: pass 1 of associate_kmers with the table row-sharded over `devices` (one entry per shard; a
    device may appear more than once) inside this process - what `associate_kmers --gpus N` runs."""

    def __init__(self, n_acc_file: int, col, Y, topn, min_count: int, devices: Sequence[int], kernel: int = capi.KERNEL_AUTO,
                 chunk_rows: int = 0, host_threads: int = 0, count_patterns: bool = False):
        self.col = np.ascontiguousarray(col, np.uint64)
        self.Y = np.ascontiguousarray(Y, np.float32)
        if self.Y.ndim == 1:
            self.Y = self.Y[None, :]
        self.n_pheno, self.n_acc = self.Y.shape
        self.topn = np.ascontiguousarray(np.broadcast_to(np.asarray(topn, np.uint64), (self.n_pheno,)))
        self.devices = np.ascontiguousarray(devices, np.int32)
        self.words_per_row = (n_acc_file + 63) // 64
        p = capi.ScanParams()
        p.struct_size = C.sizeof(capi.ScanParams)
        p.device = 0
        p.n_acc_file = n_acc_file
        p.n_acc = self.n_acc
        p.col = self.col.ctypes.data_as(C.POINTER(C.c_uint64))
        p.n_pheno = self.n_pheno
        p.Y = self.Y.ctypes.data_as(C.POINTER(C.c_float))
        p.topn = self.topn.ctypes.data_as(C.POINTER(C.c_uint64))
        p.min_count = min_count
        p.chunk_rows = chunk_rows
        p.host_threads = host_threads
        p.kernel = kernel
        p.record_history = 0
        p.count_patterns = 1 if count_patterns else 0
        self._h = C.c_void_p()
        check(lib.kgwas_multiscan_create(C.byref(p), ptr(self.devices), len(self.devices), C.byref(self._h)))

    def run_table(self, table: "KmersTable", row0: int, n_rows: int):
        check(lib.kgwas_multiscan_run_table(self._h, table._h, row0, n_rows))

    def run_device(self, d_ptrs: Sequence[int], n_rows: Sequence[int], first_rows: Sequence[int]):
        G = len(self.devices)
        pp = (C.c_void_p * G)(*[C.c_void_p(int(x)) for x in d_ptrs])
        nr = np.ascontiguousarray(n_rows, np.uint64)
        fr = np.ascontiguousarray(first_rows, np.uint64)
        check(lib.kgwas_multiscan_run_device(self._h, pp, ptr(nr), ptr(fr)))

    def finish(self):
        check(lib.kgwas_multiscan_finish(self._h))

    def result(self, j: int):
        n = C.c_uint64()
        k, s, r = C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)(), C.POINTER(C.c_uint64)()
        check(lib.kgwas_multiscan_result(self._h, j, C.byref(n), C.byref(k), C.byref(s), C.byref(r)))
        m = n.value
        if m == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.float64), np.zeros(0, np.uint64)
        return (np.ctypeslib.as_array(k, shape=(m,)).copy(), np.ctypeslib.as_array(s, shape=(m,)).copy(),
                np.ctypeslib.as_array(r, shape=(m,)).copy())

    def stats(self) -> dict:
        G = len(self.devices)
        tot = capi.ScanStats()
        per = (capi.ScanStats * G)()
        sm, mm, rs = C.c_double(), C.c_double(), C.c_uint64()
        check(lib.kgwas_multiscan_get_stats(self._h, C.byref(tot), per, C.byref(sm), C.byref(mm), C.byref(rs)))
        d = tot.as_dict()
        d.update(scan_ms=sm.value, merge_ms=mm.value, rescans=rs.value, per_shard=[x.as_dict() for x in per])
        return d

    def close(self):
        if getattr(self, "_h", None):
            lib.kgwas_multiscan_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def kinship_table_multi(table: "KmersTable", min_count: int, devices: Sequence[int]):
    """emma_kinship_kmers' accumulation over the whole table, row-sharded over `devices`: (K, n_used)."""
    dv = np.ascontiguousarray(devices, np.int32)
    H = np.zeros((table.n_acc, table.n_acc), np.uint64)
    n = C.c_uint64()
    check(lib.kgwas_kinship_table_multi(ptr(dv), len(dv), table._h, min_count, ptr(H), C.byref(n)))
    return kinship_from_partials(H, n.value), n.value


def merge_shards(topn, shard_histories, threads: int = 0):
    """Cross-shard merge. shard_histories[g][j] = (kmer, score, row) arrays of shard g (row order),
    shards listed in row order. Returns one BestAssociationsHeap per phenotype column."""
    G = len(shard_histories)
    P = len(shard_histories[0])
    topn = np.ascontiguousarray(np.broadcast_to(np.asarray(topn, np.uint64), (P,)))
    counts = np.zeros((G, P), np.uint64)
    ks, ss, rs = [], [], []
    for g in range(G):
        for j in range(P):
            counts[g, j] = len(shard_histories[g][j][0])
        ks.append(np.ascontiguousarray(np.concatenate([np.asarray(h[0], np.uint64) for h in shard_histories[g]])))
        ss.append(np.ascontiguousarray(np.concatenate([np.asarray(h[1], np.float64) for h in shard_histories[g]])))
        rs.append(np.ascontiguousarray(np.concatenate([np.asarray(h[2], np.uint64) for h in shard_histories[g]])))
    kp = (C.c_void_p * G)(*[a.ctypes.data for a in ks])
    sp = (C.c_void_p * G)(*[a.ctypes.data for a in ss])
    rp = (C.c_void_p * G)(*[a.ctypes.data for a in rs])
    out = (C.c_void_p * P)()
    check(lib.kgwas_merge_shards(P, ptr(topn), G, ptr(counts), kp, sp, rp, threads, out))
    return [BestAssociationsHeap(0, _handle=out[j]) for j in range(P)]


class Kinship:
    def __init__(self, n_acc_file: int, min_count: int, device: int = 0):
        self.n_acc = n_acc_file
        self.words_per_row = (n_acc_file + 63) // 64
        self._h = C.c_void_p()
        check(lib.kgwas_kinship_create(device, n_acc_file, min_count, C.byref(self._h)))

    def feed_host(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, np.uint64)
        check(lib.kgwas_kinship_feed_host(self._h, ptr(rows), rows.size // (1 + self.words_per_row)))

    def feed_table(self, table: "KmersTable", row0: int, n_rows: int):
        """Rows [row0, row0 + n_rows) of an open table; file read, copy and kernels of consecutive pieces overlap."""
        check(lib.kgwas_kinship_feed_table(self._h, table._h, row0, n_rows))

    def feed_device(self, d_ptr: int, n_rows: int, stream: int = 0):
        check(lib.kgwas_kinship_feed_device(self._h, C.c_void_p(d_ptr), n_rows, C.c_void_p(stream)))

    def partials(self):
        H = np.zeros((self.n_acc, self.n_acc), np.uint64)
        n = C.c_uint64()
        check(lib.kgwas_kinship_partials(self._h, ptr(H), C.byref(n)))
        return H, n.value

    def matrix(self):
        H, n = self.partials()
        return kinship_from_partials(H, n), n

    def stats(self):
        ms, l, r = C.c_double(), C.c_uint64(), C.c_uint64()
        check(lib.kgwas_kinship_get_stats(self._h, C.byref(ms), C.byref(l), C.byref(r)))
        return dict(kernel_ms=ms.value, launches=l.value, rows_fed=r.value)

    def close(self):
        if getattr(self, "_h", None):
            lib.kgwas_kinship_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def kinship_from_partials(H: np.ndarray, n_used: int) -> np.ndarray:
    H = np.ascontiguousarray(H, np.uint64)
    K = np.zeros_like(H)
    check(lib.kgwas_kinship_from_partials(H.shape[0], ptr(H), n_used, ptr(K)))
    return K


def kinship_format(K: np.ndarray, n_used: int) -> bytes:
    K = np.ascontiguousarray(K, np.uint64)
    need = lib.kgwas_kinship_format(K.shape[0], ptr(K), n_used, None, 0)
    buf = C.create_string_buffer(int(need) + 1)
    lib.kgwas_kinship_format(K.shape[0], ptr(K), n_used, buf, need)
    return buf.raw[:need]


def write_plink(out_base: str, table: KmersTable, col, acc_names, y, kmer_pop, row_pop):
    col = np.ascontiguousarray(col, np.uint64)
    y = np.ascontiguousarray(y, np.float32)
    kmer_pop = np.ascontiguousarray(kmer_pop, np.uint64)
    row_pop = np.ascontiguousarray(row_pop, np.uint64)
    arr = (C.c_char_p * len(acc_names))(*[a.encode() for a in acc_names])
    check(lib.kgwas_write_plink(out_base.encode(), table._h, ptr(col), len(col), arr, ptr(y), len(kmer_pop),
                                ptr(kmer_pop), ptr(row_pop)))


def write_plink_many(out_bases, table: KmersTable, col, acc_names, Y, kmer_pops, row_pops, threads: int = 0):
    """Pass 2 for all phenotype columns at once: column j's .bed/.bim/.fam from (kmer_pops[j], row_pops[j]) in pop order."""
    col = np.ascontiguousarray(col, np.uint64)
    Y = np.ascontiguousarray(Y, np.float32)
    n = len(out_bases)
    assert Y.shape == (n, len(col)) and len(kmer_pops) == n and len(row_pops) == n
    ks = [np.ascontiguousarray(k, np.uint64) for k in kmer_pops]
    rs = [np.ascontiguousarray(r, np.uint64) for r in row_pops]
    n_win = np.asarray([len(k) for k in ks], np.uint64)
    bases = (C.c_char_p * n)(*[b.encode() for b in out_bases])
    names = (C.c_char_p * len(acc_names))(*[a.encode() for a in acc_names])
    kp = (C.c_void_p * n)(*[a.ctypes.data for a in ks])
    rp = (C.c_void_p * n)(*[a.ctypes.data for a in rs])
    check(lib.kgwas_write_plink_many(n, bases, table._h, ptr(col), len(col), names, ptr(Y), ptr(n_win), kp, rp, threads))


def synth_rows_host(first_row: int, n_rows: int, n_acc: int, seed: int) -> np.ndarray:
    out = np.empty((n_rows, 1 + (n_acc + 63) // 64), np.uint64)
    check(lib.kgwas_synth_rows_host(ptr(out), first_row, n_rows, n_acc, seed))
    return out


def synth_rows_device(d_ptr: int, first_row: int, n_rows: int, n_acc: int, seed: int, stream: int = 0):
    check(lib.kgwas_synth_rows_device(C.c_void_p(d_ptr), first_row, n_rows, n_acc, seed, C.c_void_p(stream)))


def table_to_bed(out_base: str, table: KmersTable, col, acc_names: Sequence[str], y, min_count: int, batch_size: int,
                 unique_patterns: bool = False, device: int = 0):
    """kmers_table_to_bed: the MAC-filtered table as PLINK files <out_base>.<i>.{bed,bim,fam}, a new set after every
    batch_size kept k-mers; unique_patterns keeps the first k-mer of each presence/absence pattern.
    Returns (batches, k-mers written)."""
    col = np.ascontiguousarray(col, np.uint64)
    y = np.ascontiguousarray(y, np.float32)
    arr = (C.c_char_p * len(acc_names))(*[a.encode() for a in acc_names])
    nb, nw = C.c_uint64(0), C.c_uint64(0)
    check(lib.kgwas_table_to_bed(table._h, ptr(col), len(col), arr, ptr(y), min_count, batch_size, 1 if unique_patterns else 0,
                                 out_base.encode(), device, C.byref(nb), C.byref(nw)))
    return nb.value, nw.value


class SnpsDataBase:
    """MultipleSNPsDataBases (src/snps_multiple_databases.h:25-63): a PLINK bed/bim/fam trio restricted to the
    phenotyped samples (phenotype order); scoring runs on the GPU."""

    def __init__(self, base_bedbim: str, samples: Sequence[str]):
        self._h = C.c_void_p()
        self.samples = list(samples)
        arr = (C.c_char_p * len(self.samples))(*[a.encode() for a in self.samples])
        check(lib.kgwas_snps_open(base_bedbim.encode(), arr, len(self.samples), C.byref(self._h)))
        n, f, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.kgwas_snps_info(self._h, C.byref(n), C.byref(f), C.byref(b)))
        self.n_snps, self.n_samples_file, self.bytes_per_snp = n.value, f.value, b.value

    def scores(self, Y: np.ndarray, mac: float, device: int = 0) -> np.ndarray:
        """calculate_grammmar_approx_association of every SNP for every row of Y[n_pheno][n_samples]."""
        Y = np.ascontiguousarray(np.atleast_2d(Y), np.float32)
        out = np.zeros((Y.shape[0], self.n_snps), np.float64)
        check(lib.kgwas_snps_scores(self._h, ptr(Y), Y.shape[0], float(mac), device, ptr(out)))
        return out

    def best(self, Y: np.ndarray, topn: int, mac: float, device: int = 0):
        """get_most_associated_snps per phenotype column: list of sorted SNP index arrays."""
        Y = np.ascontiguousarray(np.atleast_2d(Y), np.float32)
        P = Y.shape[0]
        counts = np.zeros(P, np.uint64)
        idx = np.zeros((P, max(topn, 1)), np.uint64)
        check(lib.kgwas_snps_best(self._h, ptr(Y), P, topn, float(mac), device, ptr(counts), ptr(idx)))
        return [idx[j, : int(counts[j])].copy() for j in range(P)]

    def write(self, out_bases: Sequence[str], index_lists):
        """output_plink_bed_file: list l (sorted SNP indices) -> out_bases[l].bed/.bim."""
        n = len(out_bases)
        stride = max([len(x) for x in index_lists] + [1])
        counts = np.asarray([len(x) for x in index_lists], np.uint64)
        idx = np.zeros((n, stride), np.uint64)
        for l, x in enumerate(index_lists):
            idx[l, : len(x)] = x
        arr = (C.c_char_p * n)(*[b.encode() for b in out_bases])
        check(lib.kgwas_snps_write(self._h, n, arr, ptr(counts), ptr(idx), stride))

    def close(self):
        if self._h:
            lib.kgwas_snps_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
