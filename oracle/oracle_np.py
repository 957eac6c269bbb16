"""oracle/oracle_np.py — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT.

NumPy / pure-Python restatement of the voichek/kmersGWAS association-scan hot path,
independent of oracle.cpp (a third statement of the same float32 order) plus a
pure-Python restatement of libstdc++'s binary-heap algorithms so that the
std::priority_queue tie behaviour the reference inherits can be cross-checked.

PARITY STATUS: parity unpinned — see the header of oracle.cpp for why the reference
cannot be executed in this image. Citations are relative to /root/reference/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
import numpy as np

HEADER_MAGIC = 0xDDCCBBAA  # src/kmers_multiple_databases.cpp:78 / kmers_merge_multiple_databaes.cpp:54-60


# ---------------------------------------------------------------------------
# a-1  on-disk format
# ---------------------------------------------------------------------------
def write_table(base: str, names: list[str], kmer_len: int, kmers: np.ndarray, words: np.ndarray) -> None:
    """Writer of the .table/.names pair (src/kmers_merge_multiple_databaes.cpp:54-73,
    src/build_kmers_table.cpp:80-91): header = u32 magic, u64 n_accessions, u32 k;
    rows = [u64 kmer][W_f x u64], sample c is bit c%64 of word c//64."""
    n_acc = len(names)
    w_f = (n_acc + 63) // 64
    words = np.ascontiguousarray(words, dtype=np.uint64).reshape(len(kmers), w_f)
    body = np.empty((len(kmers), 1 + w_f), dtype=np.uint64)
    body[:, 0] = kmers
    body[:, 1:] = words
    with open(base + ".table", "wb") as f:
        f.write(np.uint32(HEADER_MAGIC).tobytes())
        f.write(np.uint64(n_acc).tobytes())
        f.write(np.uint32(kmer_len).tobytes())
        f.write(body.tobytes())
    with open(base + ".names", "w") as f:
        for n in names:
            f.write(n + "\n")


def read_table(base: str, kmer_len: int):
    """Reader guards of MultipleKmersDataBases' ctor (src/kmers_multiple_databases.cpp:65-93)."""
    names = open(base + ".names").read().split()  # load_kmers_talbe_column_names: whitespace tokens
    raw = np.fromfile(base + ".table", dtype=np.uint8)
    if raw.size <= 16:
        raise ValueError("Kmer table size is too small")
    magic = int(raw[0:4].view(np.uint32)[0])
    n_acc = int(raw[4:12].view(np.uint64)[0])
    k = int(raw[12:16].view(np.uint32)[0])
    if magic != HEADER_MAGIC:
        raise ValueError("Incorrect prefix")
    if n_acc != len(names):
        raise ValueError("Number of accession in file not as defined in class")
    if k != kmer_len:
        raise ValueError("Kmer length not as defined in class")
    w_f = (n_acc + 63) // 64
    body = raw[16:]
    if body.size % (8 * (1 + w_f)) != 0:
        raise ValueError("size of file not valid")
    rows = body.view(np.uint64).reshape(-1, 1 + w_f)
    return names, rows


def column_map(names_file: list[str], names_pheno: list[str]) -> np.ndarray:
    """create_map_from_all_DBs (src/kmers_multiple_databases.cpp:297-311) +
    get_index_DB's duplicate check (src/kmer_general.cpp:227-237)."""
    col = []
    for n in names_pheno:
        hits = [i for i, m in enumerate(names_file) if m == n]
        if len(hits) > 1:
            raise ValueError("Two DBs with the same name! " + n)
        if not hits:
            raise ValueError("Couldn't find path for DB: " + n)
        col.append(hits[0])
    return np.asarray(col, dtype=np.uint64)


def load_phenotypes(path: str):
    """load_phenotypes_file (src/kmer_general.cpp:175-205): TSV, header row names the
    columns, every data row must have 1 + n_pheno tab fields, values parsed as float32."""
    names, acc, vals = [], [], []
    with open(path) as f:
        for ln, line in enumerate(f.read().split("\n")):
            if line == "" and ln > 0:
                # getline stops at EOF; a trailing newline produces no extra record
                continue
            toks = line.split("\t")
            if ln == 0:
                names = toks[1:]
            else:
                if len(toks) != len(names) + 1:
                    raise ValueError("File should have the same number of fields in each row")
                acc.append(toks[0])
                from . import binding  # std::stof itself (decimal -> float32 in one rounding)
                vals.append([np.float32(binding.stof(t)) for t in toks[1:]])
    Y = np.asarray(vals, dtype=np.float32).T.copy() if vals else np.zeros((len(names), 0), np.float32)
    return names, acc, Y


def min_count(S: int, maf: float, mac: int) -> int:
    """src/associate_kmers.cpp:99-103."""
    return max(int(math.ceil(float(S) * maf)), mac)


# ---------------------------------------------------------------------------
# a-3  MAC filter + squeeze
# ---------------------------------------------------------------------------
def _popcount64(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.uint64)
    out = np.zeros(a.shape, dtype=np.uint64)
    for sh in range(0, 64, 8):
        out += _POP8[((a >> np.uint64(sh)) & np.uint64(0xFF)).astype(np.intp)]
    return out


_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint64)


def unpack_bits(rows: np.ndarray, col: np.ndarray) -> np.ndarray:
    """g[r, i] = bit col[i] of file row r (the per-bit squeeze of
    src/kmers_multiple_databases.cpp:125-132, kept unpacked)."""
    col = col.astype(np.int64)
    words = rows[:, 1:]
    return ((words[:, col // 64] >> (col % 64).astype(np.uint64)) & np.uint64(1)).astype(np.uint8)


def mac_filter(rows: np.ndarray, col: np.ndarray, mac: int):
    g = unpack_bits(rows, col)
    n1 = g.sum(axis=1).astype(np.int64)
    S = len(col)
    keep = (n1 >= mac) & (n1 <= S - mac) if S - mac >= 0 else np.zeros(len(n1), bool)
    return g, n1, keep


# ---------------------------------------------------------------------------
# a-4 / a-5  scoring
# ---------------------------------------------------------------------------
def padded_len(S: int) -> int:
    return 64 * 2 * ((S + 127) // 128)  # 64 * W_m (src/kmers_multiple_databases.cpp:51)


def permuted_sum(y: np.ndarray) -> np.float32:
    """update_scores_and_sum (src/kmers_multiple_databases.cpp:288-295): sequential float32
    sum over the padded, permuted vector R[128b+4s+l] = V[128b+32l+31-s]."""
    S = len(y)
    L = padded_len(S)
    V = np.zeros(L, dtype=np.float32)
    V[:S] = y
    acc = np.float32(0)
    for b in range(L // 128):
        for s in range(32):
            for l in range(4):
                acc = np.float32(acc + V[128 * b + 32 * l + 31 - s])
    return acc


def scores(g: np.ndarray, n1: np.ndarray, y: np.ndarray, mac: int) -> np.ndarray:
    """calculate_kmer_score (src/kmers_multiple_databases.cpp:327-363) for every row of the
    unpacked matrix g (rows x S). Vectorised over rows; the S/4 sequential float32 adds per
    SSE lane are kept in the reference order."""
    M, S = g.shape
    L = padded_len(S)
    V = np.zeros(L, dtype=np.float32)
    V[:S] = y
    G = np.zeros((M, L), dtype=bool)
    G[:, :S] = g.astype(bool)
    acc = np.zeros((4, M), dtype=np.float32)
    zero = np.float32(0)
    for b in range(L // 128):
        for s in range(32):
            for l in range(4):
                i = 128 * b + 32 * l + 31 - s
                acc[l] = (acc[l] + np.where(G[:, i], V[i], zero)).astype(np.float32)
    yf = ((acc[0] + acc[1]).astype(np.float32) + acc[2]).astype(np.float32)
    yf = (yf + acc[3]).astype(np.float32)
    yigi = yf.astype(np.float64)
    N = np.float64(S)
    N1 = n1.astype(np.float64)
    ssum = np.float64(permuted_sum(y))
    with np.errstate(all="ignore"):
        r = N * yigi - N1 * ssum
        r = r * r
        sc = r / (N * N1 - N1 * N1)
    ok = (np.float64(mac) <= (N - N1)) & (np.float64(mac) <= N1)
    return np.where(ok, sc, 0.0)


# ---------------------------------------------------------------------------
# a-7  heap: pure-Python restatement of libstdc++ (GCC 11, bits/stl_heap.h)
#      std::push_heap / std::pop_heap as std::priority_queue uses them.
# ---------------------------------------------------------------------------
class StdHeap:
    """std::priority_queue<tuple<kmer,score,row>, vector, cmp> with cmp(a,b) = a.score > b.score
    (src/kmer_general.h:113-128). libstdc++'s algorithms (third-party to the reference; GCC 11.4
    in this image and on the GPU box), restated from bits/stl_heap.h:
      __push_heap: sift the new last element up while cmp(parent, value);
      __adjust_heap: move the larger-by-cmp child up to the hole until the bottom, then __push_heap;
      pop_heap: swap-out first, __adjust_heap(0, len-1, value = old last)."""

    def __init__(self):
        self.v = []

    @staticmethod
    def _cmp(a, b):
        return a[1] > b[1]

    def _push_heap(self, hole, top, value):
        v = self.v
        parent = (hole - 1) // 2
        while hole > top and self._cmp(v[parent], value):
            v[hole] = v[parent]
            hole = parent
            parent = (hole - 1) // 2
        v[hole] = value

    def push(self, e):
        self.v.append(e)
        self._push_heap(len(self.v) - 1, 0, e)

    def top(self):
        return self.v[0]

    def pop(self):
        v = self.v
        value = v[-1]
        v[-1] = v[0]
        length = len(v) - 1
        hole, top = 0, 0
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if self._cmp(v[child], v[child - 1]):
                child -= 1
            v[hole] = v[child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            v[hole] = v[child - 1]
            hole = child - 1
        if length > 0:
            self._push_heap(hole, top, value)
        v.pop()

    def __len__(self):
        return len(self.v)


class BestHeap:
    """BestAssociationsHeap (src/best_associations_heap.cpp:26-59, 82-127)."""

    def __init__(self, n):
        self.n = n
        self.q = StdHeap()
        self.cnt = 0
        self.lowest = 0.0

    def add(self, kmer, score, row):
        self.cnt += 1
        if len(self.q) < self.n:
            self.q.push((kmer, score, row))
            self.lowest = self.q.top()[1]
        elif score > self.lowest:
            self.q.pop()
            self.q.push((kmer, score, row))
            self.lowest = self.q.top()[1]

    def pop_all(self):
        tmp = StdHeap()
        tmp.v = list(self.q.v)
        out = []
        while len(tmp):
            out.append(tmp.top())
            tmp.pop()
        return out

    def output_list(self):
        pops = self.pop_all()
        n = len(pops)
        lst = [(k, n - i, row) for i, (k, _, row) in enumerate(pops)]
        lst.sort(key=lambda t: t[2])
        return lst


# ---------------------------------------------------------------------------
# a-9  kinship
# ---------------------------------------------------------------------------
def kinship(rows: np.ndarray, S_f: int, maf: float):
    """emma_kinship_kmers (src/emma_kinship_kmers.cpp:77-102) with
    update_emma_kinshhip_calculation (src/kmers_multiple_databases.cpp:418-438),
    through the closed form K_ij = sum_rows (1 ^ g_i ^ g_j) = n - hamming(i, j)."""
    mc = int(math.ceil(float(S_f) * maf))
    g = unpack_bits(rows, np.arange(S_f, dtype=np.uint64)).astype(np.int64)
    n1 = g.sum(axis=1)
    keep = (n1 >= mc) & (n1 <= S_f - mc)
    g = g[keep]
    n = int(keep.sum())
    c = g.T @ g
    ci = np.diag(c)
    K = n - ci[:, None] - ci[None, :] + 2 * c
    K = np.tril(K, -1).astype(np.uint64)
    return K, n


# ---------------------------------------------------------------------------
# misc
# ---------------------------------------------------------------------------
def bits2kmer(w: int, k: int) -> str:
    """bits2kmer31 (src/kmer_general.cpp:77-87)."""
    s = []
    for _ in range(k):
        s.append("ACGT"[w & 3])
        w >>= 2
    return "".join(reversed(s))
