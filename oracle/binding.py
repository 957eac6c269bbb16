"""oracle/binding.py — ctypes view of liboracle.so. TEST INFRASTRUCTURE ONLY (see oracle.cpp header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build() -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_prepare_scores.restype = C.c_float
        L.orc_prepare_scores.argtypes = [f32p, C.c_uint64, C.c_uint64, f32p]
        L.orc_score.restype = C.c_double
        L.orc_score.argtypes = [u64p, f32p, C.c_float, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64, C.c_int]
        L.orc_squeeze.restype = C.c_uint64
        L.orc_squeeze.argtypes = [u64p, C.c_uint64, C.c_uint64, u64p, C.c_uint64, C.c_uint64, u64p, u32p, u8p]
        L.orc_scores_dense.restype = C.c_uint64
        L.orc_scores_dense.argtypes = [u64p, C.c_uint64, C.c_uint64, u64p, C.c_uint64, f32p, C.c_uint64,
                                       C.c_uint64, C.c_int, f64p, u8p]
        L.orc_heap_new.restype = C.c_void_p
        L.orc_heap_new.argtypes = [C.c_uint64]
        L.orc_heap_free.argtypes = [C.c_void_p]
        L.orc_heap_add.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_uint64]
        L.orc_heap_add_many.argtypes = [C.c_void_p, u64p, f64p, u64p, C.c_uint64]
        L.orc_heap_size.restype = C.c_uint64
        L.orc_heap_size.argtypes = [C.c_void_p]
        L.orc_heap_insertions.restype = C.c_uint64
        L.orc_heap_insertions.argtypes = [C.c_void_p]
        L.orc_heap_lowest.restype = C.c_double
        L.orc_heap_lowest.argtypes = [C.c_void_p]
        L.orc_heap_pop_all.argtypes = [C.c_void_p, u64p, f64p, u64p]
        L.orc_heap_output_list.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.orc_associate.restype = C.c_uint64
        L.orc_associate.argtypes = [u64p, C.c_uint64, C.c_uint64, u64p, C.c_uint64, f32p, C.c_uint64, u64p,
                                    C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, f64p, u64p, u64p,
                                    C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.orc_write_plink.restype = C.c_int
        L.orc_write_plink.argtypes = [C.c_char_p, u64p, C.c_uint64, C.c_uint64, u64p, C.c_uint64,
                                      C.POINTER(C.c_char_p), f32p, C.c_uint64, C.c_uint64, u64p, u64p]
        L.orc_kinship.restype = C.c_uint64
        L.orc_kinship.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_uint64, u64p]
        L.orc_kinship_text.restype = C.c_uint64
        L.orc_kinship_text.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]
        L.orc_stof.restype = C.c_float
        L.orc_stof.argtypes = [C.c_char_p]
        L.orc_bits2kmer.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p]
        L.orc_pattern_hash.restype = C.c_uint64
        L.orc_pattern_hash.argtypes = [u64p, C.c_uint64]
        L.orc_min_count.restype = C.c_uint64
        L.orc_min_count.argtypes = [C.c_uint64, C.c_double, C.c_uint64]
        _LIB = L
    return _LIB


def _rows(rows):
    return np.ascontiguousarray(rows, dtype=np.uint64)


def scores_dense(rows, S_f, col, Y, mac, which=1):
    """(n_pheno x n_rows float64 scores, kept mask)."""
    rows = _rows(rows)
    n_rows = rows.shape[0]
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    col = np.ascontiguousarray(col, dtype=np.uint64)
    out = np.zeros((Y.shape[0], n_rows), dtype=np.float64)
    kept = np.zeros(n_rows, dtype=np.uint8)
    lib().orc_scores_dense(rows.reshape(-1), n_rows, S_f, col, len(col), Y.reshape(-1), Y.shape[0], mac, which,
                           out.reshape(-1), kept)
    return out, kept.astype(bool)


def associate(rows, S_f, col, Y, topn, mac, batch_size=10_000_000, threads=1, count_patterns=False):
    """Pass 1 of associate_kmers. Returns dict with per-phenotype pop-order lists."""
    rows = _rows(rows)
    n_rows = rows.shape[0]
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    col = np.ascontiguousarray(col, dtype=np.uint64)
    P = Y.shape[0]
    topn = np.ascontiguousarray(np.broadcast_to(np.asarray(topn, dtype=np.uint64), (P,)))
    cap = int(max(1, min(int(topn.max()), n_rows)))
    out_n = np.zeros(P, np.uint64)
    out_kmer = np.zeros(P * cap, np.uint64)
    out_score = np.zeros(P * cap, np.float64)
    out_ref = np.zeros(P * cap, np.uint64)
    out_file = np.zeros(P * cap, np.uint64)
    npat = C.c_uint64(0)
    pushes = C.c_uint64(0)
    timing = (C.c_double * 2)()
    tested = lib().orc_associate(rows.reshape(-1), n_rows, S_f, col, len(col), Y.reshape(-1), P, topn, mac,
                                 batch_size, threads, cap, out_n, out_kmer, out_score, out_ref, out_file,
                                 1 if count_patterns else 0, C.byref(npat), timing, C.byref(pushes))
    res = []
    for j in range(P):
        n = int(out_n[j])
        sl = slice(j * cap, j * cap + n)
        res.append(dict(kmer=out_kmer[sl].copy(), score=out_score[sl].copy(), ref_row=out_ref[sl].copy(),
                        file_row=out_file[sl].copy()))
    return dict(tested=int(tested), per_pheno=res, patterns=int(npat.value), pushes=int(pushes.value),
                t_load=timing[0], t_score=timing[1])


def kinship(rows, S_f, min_count):
    rows = _rows(rows)
    K = np.zeros((S_f, S_f), dtype=np.uint64)
    n = lib().orc_kinship(rows.reshape(-1), rows.shape[0], S_f, min_count, K.reshape(-1))
    return K, int(n)


def kinship_text(K, n):
    K = np.ascontiguousarray(K, dtype=np.uint64)
    need = lib().orc_kinship_text(K.reshape(-1), K.shape[0], n, None, 0)
    buf = C.create_string_buffer(int(need))
    lib().orc_kinship_text(K.reshape(-1), K.shape[0], n, buf, need)
    return buf.raw[:need]


def write_plink(base, rows, S_f, col, acc_names, y, kmer_len, kmer_pop, filerow_pop):
    rows = _rows(rows)
    col = np.ascontiguousarray(col, dtype=np.uint64)
    arr = (C.c_char_p * len(acc_names))(*[a.encode() for a in acc_names])
    kmer_pop = np.ascontiguousarray(kmer_pop, np.uint64)
    filerow_pop = np.ascontiguousarray(filerow_pop, np.uint64)
    rc = lib().orc_write_plink(base.encode(), rows.reshape(-1), rows.shape[0], S_f, col, len(col), arr,
                               np.ascontiguousarray(y, np.float32), kmer_len, len(kmer_pop), kmer_pop, filerow_pop)
    if rc != 0:
        raise RuntimeError("orc_write_plink failed: %d" % rc)


def table_to_bed(base, rows, S_f, col, acc_names, y, kmer_len, min_count, batch_size, unique):
    """kmers_table_to_bed on in-memory rows; writes <base>.<batch>.{bed,bim,fam}; returns (batches, k-mers written)."""
    rows = _rows(rows)
    col = np.ascontiguousarray(col, dtype=np.uint64)
    arr = (C.c_char_p * len(acc_names))(*[a.encode() for a in acc_names])
    nw = C.c_uint64(0)
    L = lib()
    L.orc_table_to_bed.restype = C.c_uint64
    L.orc_table_to_bed.argtypes = [C.c_char_p, np.ctypeslib.ndpointer(np.uint64, flags="C"), C.c_uint64, C.c_uint64,
                                   np.ctypeslib.ndpointer(np.uint64, flags="C"), C.c_uint64, C.c_void_p,
                                   np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                   C.POINTER(C.c_uint64)]
    nb = L.orc_table_to_bed(base.encode(), rows.reshape(-1), rows.shape[0], S_f, col, len(col), arr,
                            np.ascontiguousarray(y, np.float32), kmer_len, min_count, batch_size, 1 if unique else 0, C.byref(nw))
    return int(nb), int(nw.value)


def snps_scores(bed_body, n_samples_file, sample_index, y, mac):
    """calculate_grammmar_approx_association of every SNP of a .bed body (bytes after the magic) for the samples
    whose positions in the .fam are sample_index (phenotype order) and their phenotype y."""
    bed_body = np.ascontiguousarray(np.frombuffer(bed_body, np.uint8))
    bps = (n_samples_file + 3) // 4
    n_snps = len(bed_body) // bps
    idx = np.asarray(sample_index, np.uint64)
    byte_idx = np.ascontiguousarray(idx // 4, np.uint64)
    shift = np.ascontiguousarray((idx % 4) * 2, np.uint64)
    out = np.zeros(n_snps, np.float64)
    L = lib()
    L.orc_snps_scores.restype = None
    L.orc_snps_scores.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C"), C.c_uint64, C.c_uint64,
                                  np.ctypeslib.ndpointer(np.uint64, flags="C"), np.ctypeslib.ndpointer(np.uint64, flags="C"),
                                  C.c_uint64, np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_double,
                                  np.ctypeslib.ndpointer(np.float64, flags="C")]
    L.orc_snps_scores(bed_body, n_snps, bps, byte_idx, shift, len(idx), np.ascontiguousarray(y, np.float32), float(mac), out)
    return out


class Heap:
    def __init__(self, n):
        self.h = lib().orc_heap_new(n)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_heap_free(self.h)
            self.h = None

    def add(self, kmer, score, row):
        lib().orc_heap_add(self.h, int(kmer), float(score), int(row))

    def add_many(self, kmer, score, row):
        lib().orc_heap_add_many(self.h, np.ascontiguousarray(kmer, np.uint64), np.ascontiguousarray(score, np.float64),
                                np.ascontiguousarray(row, np.uint64), len(kmer))

    def pop_all(self):
        n = int(lib().orc_heap_size(self.h))
        k = np.zeros(n, np.uint64)
        s = np.zeros(n, np.float64)
        r = np.zeros(n, np.uint64)
        if n:
            lib().orc_heap_pop_all(self.h, k, s, r)
        return k, s, r

    def output_list(self):
        n = int(lib().orc_heap_size(self.h))
        k = np.zeros(n, np.uint64)
        rk = np.zeros(n, np.uint64)
        r = np.zeros(n, np.uint64)
        if n:
            lib().orc_heap_output_list(self.h, k, rk, r)
        return k, rk, r

    @property
    def insertions(self):
        return int(lib().orc_heap_insertions(self.h))

    @property
    def lowest(self):
        return float(lib().orc_heap_lowest(self.h))


def stof(s: str) -> float:
    return float(lib().orc_stof(s.encode()))


def bits2kmer(w: int, k: int) -> str:
    buf = C.create_string_buffer(64)
    lib().orc_bits2kmer(w, k, buf)
    return buf.value.decode()
