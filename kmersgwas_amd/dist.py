"""Multi-GPU plumbing for the association scan and the kinship accumulation.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests). The k-mer table is row-sharded into contiguous ranges, every rank scans its own
shard with no data-path collective, and one small exchange closes the job:

  association : each rank's heap-push history (the effective add_association calls, row order) is
                gathered to rank 0, which replays shard 0, 1, ... in order through fresh heaps
                (kgwas_merge_shards). A shard-local heap minimum is a valid lower bound of the global
                one, so the union of histories contains every globally effective push.
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
    shard_histories[g][j] for g = 0..world-1 (None elsewhere). Three all_gathers of flat tensors."""
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


def merge_on_root(scan: "engine.AssociationScan", topn, dst: int = 0, threads: int = 0):
    """Gather every rank's history and replay on `dst`. Returns (heaps, tested_total) on dst."""
    P = scan.n_pheno
    hist = [scan.history(j) for j in range(P)]
    tested = torch.tensor([scan.stats()["rows_tested"]], dtype=torch.int64, device=_dev())
    dist.all_reduce(tested, op=dist.ReduceOp.SUM)
    shards = gather_histories(hist, dst)
    if dist.get_rank() != dst:
        return None, int(tested.item())
    return engine.merge_shards(topn, shards, threads), int(tested.item())


def allreduce_kinship(H: np.ndarray, n_used: int):
    """Sum the integer Hamming partials and used-row counts over ranks."""
    dev = _dev()
    t = torch.from_numpy(np.ascontiguousarray(H, np.uint64).view(np.int64)).to(dev)
    n = torch.tensor([n_used], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.uint64), int(n.item())
