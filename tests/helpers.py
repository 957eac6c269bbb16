"""Shared generators for the tests (test infrastructure, not product)."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def random_table(n_rows, n_acc, seed, dup_frac=0.0, freq_lo=0.02, freq_hi=0.98):
    """Random .table body: kmers ascending, per-row frequency uniform in [lo, hi]; optionally a
    fraction of rows duplicates an earlier row's presence/absence pattern (-> tied scores)."""
    rng = np.random.default_rng(seed)
    W = (n_acc + 63) // 64
    f = rng.uniform(freq_lo, freq_hi, size=n_rows)
    bits = rng.random((n_rows, n_acc)) < f[:, None]
    if dup_frac > 0 and n_rows > 1:
        n_dup = int(n_rows * dup_frac)
        dst = rng.choice(np.arange(1, n_rows), size=n_dup, replace=False)
        src = (rng.random(n_dup) * dst).astype(np.int64)
        bits[dst] = bits[src]
    pad = np.zeros((n_rows, W * 64), dtype=bool)
    pad[:, :n_acc] = bits
    words = np.packbits(pad.reshape(n_rows, W, 64), axis=2, bitorder="little").view(np.uint64).reshape(n_rows, W)
    rows = np.empty((n_rows, 1 + W), dtype=np.uint64)
    rows[:, 0] = np.sort(rng.choice(1 << 40, size=n_rows, replace=False)).astype(np.uint64)
    rows[:, 1:] = words
    return rows


def phenotypes(n_acc, n_perm, seed, binary=False):
    """Column 0 ~ N(0,1) (or a 0/1 trait), columns 1..n_perm = permutations of column 0."""
    rng = np.random.default_rng(seed)
    if binary:
        y0 = (rng.random(n_acc) < 0.78).astype(np.float32)
    else:
        y0 = rng.standard_normal(n_acc).astype(np.float32)
    Y = [y0]
    for _ in range(n_perm):
        Y.append(rng.permutation(y0))
    return np.ascontiguousarray(np.stack(Y).astype(np.float32))


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_rows_numpy(first_row, n_rows, n_acc, seed):
    """NumPy statement of kmersgwas_amd/csrc/synth.h (checks the host twin and the device kernel)."""
    W = (n_acc + 63) // 64
    r = np.arange(first_row, first_row + n_rows, dtype=np.uint64)
    seed = np.uint64(seed)
    with np.errstate(over="ignore"):
        q = np.uint64(5) + splitmix64(seed ^ (r * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x2545F4914F6CDD1D))) % np.uint64(246)
        h = splitmix64(seed + r * np.uint64(0x9E3779B97F4A7C15))
        out = np.empty((n_rows, 1 + W), dtype=np.uint64)
        out[:, 0] = r + np.uint64(1)
        for w in range(1, W + 1):
            acc = np.zeros(n_rows, dtype=np.uint64)
            for i in range(8):
                rnd = splitmix64(h ^ (np.uint64(w * 8 + i + 1) * np.uint64(0xC2B2AE3D27D4EB4F)))
                bit = ((q >> np.uint64(i)) & np.uint64(1)).astype(bool)
                acc = np.where(bit, acc | rnd, acc & rnd)
            if w == W and (n_acc & 63):
                acc &= np.uint64((1 << (n_acc & 63)) - 1)
            out[:, w] = acc
    return out
