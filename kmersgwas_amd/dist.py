"""Multi-GPU plumbing for the association scan and the kinship accumulation.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests). The k-mer table is row-sharded into contiguous ranges, every rank scans its own
shard with no data-path collective, and one small exchange closes the job:

  association : rank 0's heaps after its own shard ARE the global heaps after those rows. Every later
                rank g sends its heap-push history (the effective add_association calls, row order)
                filtered by  score > max(final heap minima of the full heaps of shards < g)  — anything
                else is rejected by add_association whenever it arrives, since the global minimum at
                that point is at least that large. Rank 0 replays shard 1, 2, ... in order into its own
                heaps (kgwas_scan_absorb). Volume per rank is about top-N entries per column instead of
                N*(1+ln(rows/N)).
  kinship     : integer Hamming partials + used-row counts are all-reduced (sum).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import engine


def shard_range(n_rows: int, rank: int, world: int):
    """Contiguous row range of `rank` (global row order == concatenation of shards)."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_histories(histories, dst: int = 0):
    """histories[j] = (kmer u64, score f64, row u64) of this rank. Returns on rank `dst` the list
    shard_histories[g][j] for g = 0..world-1 (None elsewhere). Four all_gathers of flat tensors."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _dev()
    P = len(histories)
    counts = torch.tensor([len(h[0]) for h in histories], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros(P, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = torch.stack(all_counts).cpu().numpy()
    total = all_counts.sum(axis=1)
    pad = int(total.max()) if total.size else 0

    def flat(idx, dtype):
        if P == 0 or total[rank] == 0:
            a = np.zeros(0, dtype)
        else:
            a = np.concatenate([np.asarray(h[idx], dtype) for h in histories])
        out = np.zeros(pad, dtype)
        out[: len(a)] = a
        return out

    gathered = []
    for idx, dt in ((0, np.uint64), (1, np.float64), (2, np.uint64)):
        mine = torch.from_numpy(flat(idx, dt).view(np.int64 if dt == np.uint64 else np.float64)).to(dev)
        bufs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(bufs, mine)
        gathered.append([b.cpu().numpy() for b in bufs] if rank == dst else None)
    if rank != dst:
        return None
    shards = []
    for g in range(world):
        off = np.concatenate([[0], np.cumsum(all_counts[g])]).astype(np.int64)
        k = gathered[0][g].view(np.uint64)
        s = gathered[1][g]
        r = gathered[2][g].view(np.uint64)
        shards.append([(k[off[j]:off[j + 1]], s[off[j]:off[j + 1]], r[off[j]:off[j + 1]]) for j in range(P)])
    return shards


def prefix_thresholds(lowest: np.ndarray, full: np.ndarray) -> np.ndarray:
    """thr[g][j] = max over shards h < g with a full heap j of lowest[h][j]; -inf if none (no filtering)."""
    G, P = lowest.shape
    thr = np.full((G, P), -np.inf)
    run = np.full(P, -np.inf)
    for g in range(G):
        thr[g] = run
        cand = np.where(full[g], lowest[g], -np.inf)
        run = np.fmax(run, cand)  # fmax ignores NaN: a NaN minimum gives no bound
    return thr


def filter_history(hist, thr_row):
    """Drop entries add_association is guaranteed to reject (score not above thr). NaN scores go too:
    `score > lowest` is false for them once a heap is full, and thr > -inf means it is."""
    out = []
    for j, (k, s, r) in enumerate(hist):
        t = thr_row[j]
        if t == -np.inf:
            out.append((k, s, r))
        else:
            m = s > t
            out.append((k[m], s[m], r[m]))
    return out


def exchange_minima(low: np.ndarray, full: np.ndarray):
    """All-gather every rank's final heap minima / fullness. Returns (lowest[G][P], full[G][P])."""
    world = dist.get_world_size()
    dev = _dev()
    t = torch.from_numpy(np.concatenate([low, full.astype(np.float64)])).to(dev)
    bufs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    a = torch.stack(bufs).cpu().numpy()
    P = len(low)
    return a[:, :P], a[:, P:] > 0.5


def merge_on_root(scan: "engine.AssociationScan", dst: int = 0):
    """After every rank finished its shard scan (record_history=True): fold shards 1..G-1 into rank 0's
    heaps. Returns the total tested-k-mers count; rank 0's `scan` then holds the global result
    (call scan.finish() / scan.result(j))."""
    assert dst == 0, "the merge starts from the heaps of the first shard"
    rank = dist.get_rank()
    P = scan.n_pheno
    tested = torch.tensor([scan.stats()["rows_tested"]], dtype=torch.int64, device=_dev())
    dist.all_reduce(tested, op=dist.ReduceOp.SUM)
    low, full = scan.lowest()
    lows, fulls = exchange_minima(low, full)
    thr = prefix_thresholds(lows, fulls)
    if rank == 0:
        hist = [(np.zeros(0, np.uint64), np.zeros(0, np.float64), np.zeros(0, np.uint64))] * P  # stays local
    else:
        hist = filter_history([scan.history(j) for j in range(P)], thr[rank])
    shards = gather_histories(hist, dst)
    if rank == dst:
        scan.absorb(shards[1:])
        scan.finish()
    return int(tested.item())


def allreduce_kinship(H: np.ndarray, n_used: int):
    """Sum the integer Hamming partials and used-row counts over ranks."""
    dev = _dev()
    t = torch.from_numpy(np.ascontiguousarray(H, np.uint64).view(np.int64)).to(dev)
    n = torch.tensor([n_used], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.uint64), int(n.item())
