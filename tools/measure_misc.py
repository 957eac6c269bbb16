#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the headline bench):
  * 1-phenotype-column scan (BASELINE.md config 2', the HBM-roofline configuration) on the VALU and MFMA kernels
  * 201-column x 2048-sample scan (config 4's per-GPU shape, reduced rows)
  * kinship on a 1135-sample table (config 5's shape, reduced rows)
Usage: python tools/measure_misc.py [rows_assoc] [rows_kin]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmersgwas_amd as kg  # noqa: E402


def table(rows, S, seed=1):
    W = 1 + (S + 63) // 64
    t = torch.empty(rows * W, dtype=torch.int64, device="cuda")
    kg.synth_rows_device(t.data_ptr(), 0, rows, S, seed, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return t, W


def phen(S, P, seed=7):
    rng = np.random.default_rng(seed)
    y0 = rng.standard_normal(S).astype(np.float32)
    return np.stack([y0] + [rng.permutation(y0) for _ in range(P - 1)]).astype(np.float32)


def assoc(rows, S, P, kernel, topn=10001, reps=2):
    t, W = table(rows, S)
    scan = kg.AssociationScan(S, np.arange(S), phen(S, P), topn, kg.min_count(S, 0.05, 5), kernel=kernel)
    best = None
    for _ in range(reps + 1):
        scan.reset()
        t0 = time.perf_counter()
        scan.feed_device(t.data_ptr(), rows, 0, torch.cuda.current_stream().cuda_stream)
        scan.finish()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = scan.stats()
    kms = st["score_kernel_ms"]
    print("assoc rows=%d S=%d P=%d kernel=%s: %.1f ms/pass (kernel %.1f ms)  %.2f G rows/s  table %.0f GB/s (kernel-only %.0f GB/s = %.1f%% of 8 TB/s)  %.1f TFLOP/s useful"
          % (rows, S, P, {1: "valu", 2: "mfma", 3: "coarse"}.get(st["kernel_used"], "?"), best * 1e3, kms, rows / best / 1e9, rows * 8 * W / best / 1e9,
             rows * 8 * W / (kms * 1e-3) / 1e9, rows * 8 * W / (kms * 1e-3) / 8e12 * 100, 2.0 * rows * S * P / (kms * 1e-3) / 1e12))
    scan.close()
    del t
    torch.cuda.empty_cache()


def kinship(rows, S):
    t, W = table(rows, S)
    mc = int(np.ceil(S * 0.05))
    kin = kg.Kinship(S, mc)
    t0 = time.perf_counter()
    kin.feed_device(t.data_ptr(), rows, torch.cuda.current_stream().cuda_stream)
    K, n = kin.matrix()
    dt = time.perf_counter() - t0
    st = kin.stats()
    print("kinship rows=%d S=%d: %.1f ms total (kernels %.1f ms) -> %.3f G rows/s, %.2f T pair-updates/s, n_used=%d"
          % (rows, S, dt * 1e3, st["kernel_ms"], rows / (st["kernel_ms"] * 1e-3) / 1e9, rows * S * (S - 1) / 2 / (st["kernel_ms"] * 1e-3) / 1e12, n))
    kin.close()


if __name__ == "__main__":
    ra = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    rk = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
    assoc(ra, 1024, 1, kg.KERNEL_VALU)
    assoc(ra, 1024, 1, kg.KERNEL_MFMA)
    assoc(2 * ra, 1024, 1, kg.KERNEL_AUTO)
    assoc(ra // 5, 2048, 201, kg.KERNEL_AUTO)
    assoc(ra, 2048, 201, kg.KERNEL_AUTO)
    assoc(ra // 2, 1135, 101, kg.KERNEL_AUTO)
    kinship(rk, 1135)
    kinship(rk // 2, 241)
