"""Multi-GPU plumbing for the association scan and the kinship accumulation.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests). The k-mer table is row-sharded into contiguous ranges, every rank scans its own
shard with no data-path collective, and one small exchange closes the job:

  association : rank 0's heaps after its own shard ARE the global heaps after those rows. Every later
                rank g contributes its heap-push history (the effective add_association calls, row order)
                filtered by  score > max(final heap minima of the full heaps of shards < g)  — anything
                else is rejected by add_association whenever it arrives, since the global minimum at
                that point is at least that large: about top-N entries per column instead of
                N*(1+ln(rows/N)). merge_by_column finishes column j on rank j mod G (rank 0's heap state for
                it travels there, layout included; shards 1, 2, ... are replayed in order; the final states
                return to rank 0): one all_to_all each way, work per rank P/G columns x G shards.
                merge_to_root replays everything on rank 0 (one exchange, no heap state travels: cheaper for
                few ranks); merge_shards picks between the two. merge_on_root is the first, simple variant.
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


def _all_to_all_i64(send_parts):
    """send_parts[d] = 1-D int64 numpy array for rank d. Returns the list of arrays received from every rank."""
    world = dist.get_world_size()
    dev = _dev()
    # every message carries its length in front: no zero-sized sends or receives reach the backend
    send_parts = [np.concatenate([np.asarray([len(a)], np.int64), np.asarray(a, np.int64)]) for a in send_parts]
    ins = [int(len(a)) for a in send_parts]
    t_in = torch.tensor(ins, dtype=torch.int64, device=dev)
    all_ins = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_ins, t_in)
    rank = dist.get_rank()
    outs = [int(all_ins[src][rank]) for src in range(world)]
    flat = np.concatenate(send_parts)
    inp = torch.from_numpy(np.ascontiguousarray(flat, np.int64)).to(dev)
    out = torch.zeros(sum(outs), dtype=torch.int64, device=dev)
    dist.all_to_all_single(out, inp, outs, ins)
    out = out.cpu().numpy()
    off = np.concatenate([[0], np.cumsum(outs)]).astype(np.int64)
    msgs = [out[off[i]:off[i + 1]] for i in range(world)]
    assert all(int(m[0]) == len(m) - 1 for m in msgs)
    return [m[1:] for m in msgs]


def _pack(counts, kmer, score, row):
    """One int64 message: [counts..., kmer..., score bits..., row...]."""
    return np.concatenate([np.asarray(counts, np.uint64).view(np.int64), np.asarray(kmer, np.uint64).view(np.int64),
                           np.asarray(score, np.float64).view(np.int64), np.asarray(row, np.uint64).view(np.int64)])


def _unpack(msg, n_counts):
    counts = msg[:n_counts].view(np.uint64)
    n = int(counts.sum())
    body = msg[n_counts:]
    assert len(body) == 3 * n, (len(body), n)
    return counts, body[:n].view(np.uint64), body[n:2 * n].view(np.float64), body[2 * n:].view(np.uint64)


def merge_by_column(scan, dst: int = 0):
    """Column-distributed merge of the shard scans (every rank scanned its contiguous row range; ranks >= 1 with
    record_history=True). Column j is finished on rank j mod G: rank 0 ships that heap's state there (layout
    included), every later rank ships the part of its history that can still matter (score above the larger of
    the earlier shards' final minima), the owner replays shards 1, 2, ... in order - exactly what a single heap
    would have seen - and the final heap states return to rank 0, whose session then holds the global result
    (scan.result(j) after this call). One all_to_all each way; per rank the work is P/G columns x G shards, so
    the merge shrinks with G instead of piling up on rank 0. Returns the total tested-k-mers count.
    `scan` needs: n_pheno, stats(), lowest(), history_above(), heaps_export(), heaps_import(), absorb_flat(),
    finish()."""
    assert dst == 0, "rank 0 holds the heaps of the first shard"
    import os, time
    trace = bool(os.environ.get("KGWAS_TRACE"))
    marks = [("start", time.perf_counter())]
    mark = (lambda name: marks.append((name, time.perf_counter()))) if trace else (lambda name: None)
    rank, world = dist.get_rank(), dist.get_world_size()
    P = scan.n_pheno
    tested = torch.tensor([scan.stats()["rows_tested"]], dtype=torch.int64, device=_dev())
    dist.all_reduce(tested, op=dist.ReduceOp.SUM)
    low, full = scan.lowest()
    lows, fulls = exchange_minima(low, full)
    thr = prefix_thresholds(lows, fulls)
    owned = [np.arange(d, P, world, dtype=np.uint64) for d in range(world)]
    mark("minima")

    # way out: heap states (from rank 0) / filtered histories (from ranks >= 1), split by owner
    parts = []
    if rank == 0:
        for d in range(world):
            if d == 0 or len(owned[d]) == 0:
                parts.append(np.zeros(0, np.int64))
            else:
                parts.append(_pack(*scan.heaps_export(owned[d])))
    else:
        counts, k, s, r = scan.history_above(thr[rank])
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        for d in range(world):
            cols = owned[d].astype(np.int64)
            if len(cols) == 0:
                parts.append(np.zeros(0, np.int64))
                continue
            cat = lambda a: np.concatenate([a[off[j]:off[j + 1]] for j in cols])
            parts.append(_pack(counts[cols], cat(k), cat(s), cat(r)))
    mark("export")
    recv = _all_to_all_i64(parts)
    mark("all_to_all")

    mine = owned[rank]
    if len(mine):
        if rank != 0:
            sizes, k, s, r = _unpack(recv[0], len(mine))
            scan.heaps_import(mine, sizes, k, s, r)
        if world > 1:
            counts = np.zeros((world - 1, P), np.uint64)
            ks, ss, rs = [], [], []
            for g in range(1, world):
                c, k, s, r = _unpack(recv[g], len(mine))
                counts[g - 1, mine.astype(np.int64)] = c
                ks.append(k); ss.append(s); rs.append(r)
            scan.absorb_flat(counts, ks, ss, rs)  # entries are ordered by column, as counts says
    mark("absorb")

    # way back: final heap states of the owned columns to rank 0
    back = [np.zeros(0, np.int64) for _ in range(world)]
    if rank != 0 and len(mine):
        back[0] = _pack(*scan.heaps_export(mine))
    recv = _all_to_all_i64(back)
    if rank == 0:
        for g in range(1, world):
            if len(owned[g]):
                sizes, k, s, r = _unpack(recv[g], len(owned[g]))
                scan.heaps_import(owned[g], sizes, k, s, r)
        mark("collect")
        scan.finish()
        mark("finish")
    if trace:
        import sys
        sys.stderr.write("[kgwas] merge_by_column rank %d: %s\n" % (rank, "  ".join(
            "%s %.1f ms" % (b[0], (b[1] - a[1]) * 1e3) for a, b in zip(marks, marks[1:]))))
    return int(tested.item())


def merge_to_root(scan, dst: int = 0):
    """The same result as merge_by_column with every column finished on rank 0: ranks >= 1 send the part of their
    history that can still matter (score above the larger of the earlier shards' final minima; filtered and laid out
    flat by the library), rank 0 replays shards 1, 2, ... in order into its own heaps. One exchange, no heap state
    travels; rank 0 does P x N x ln(G) pushes (about a fifth of its own scan's at G = 8), so this is the cheaper
    variant for few ranks, merge_by_column (work per rank P/G columns, but two exchanges of the heap states) for
    many. Returns the total tested-k-mers count. `scan` needs: n_pheno, stats(), lowest(), history_above(),
    absorb_flat(), finish()."""
    assert dst == 0, "rank 0 holds the heaps of the first shard"
    rank, world = dist.get_rank(), dist.get_world_size()
    P = scan.n_pheno
    tested = torch.tensor([scan.stats()["rows_tested"]], dtype=torch.int64, device=_dev())
    dist.all_reduce(tested, op=dist.ReduceOp.SUM)
    low, full = scan.lowest()
    lows, fulls = exchange_minima(low, full)
    thr = prefix_thresholds(lows, fulls)
    parts = [np.zeros(0, np.int64) for _ in range(world)]
    if rank != 0:
        parts[0] = _pack(*scan.history_above(thr[rank]))
    recv = _all_to_all_i64(parts)
    if rank == 0:
        if world > 1:
            counts = np.zeros((world - 1, P), np.uint64)
            ks, ss, rs = [], [], []
            for g in range(1, world):
                c, k, s, r = _unpack(recv[g], P)
                counts[g - 1] = c
                ks.append(k); ss.append(s); rs.append(r)
            scan.absorb_flat(counts, ks, ss, rs)
        scan.finish()
    return int(tested.item())


def merge_shards(scan, dst: int = 0):
    """merge_to_root for up to four ranks, merge_by_column beyond (see there)."""
    return merge_to_root(scan, dst) if dist.get_world_size() <= 4 else merge_by_column(scan, dst)


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
