"""Multi-GPU plumbing for the association scan and the kinship accumulation.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests). The k-mer table is row-sharded into contiguous ranges, every rank scans its own
shard with no data-path collective, and one small exchange closes the job:

  association : rank 0's heaps after its own shard ARE the global heaps after those rows. Every later
                rank g contributes its heap-push history (the effective add_association calls, row order)
                filtered by  score > max(final heap minima of the full heaps of shards < g)  — anything
                else is rejected by add_association whenever it arrives, since the global minimum at
                that point is at least that large: about top-N entries per column instead of
                N*(1+ln(rows/N)). merge_by_column finishes the d-th block of columns on rank d (rank 0's heap
                states for it travel there, layout included; shards 1, 2, ... are replayed in order; the final
                states return to rank 0): one all_to_all each way, work per rank P/G columns x G shards; the
                messages are written by the library straight into pinned staging buffers.
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


# ---- merge messages -----------------------------------------------------------------------------------------
# A message is a run of 64-bit words: [words that follow][counts: n_cols][kmer: T][score bit patterns: T][row: T]
# (T = sum of the counts; "no columns" is the one word 0). The scan session writes its messages straight into a
# pinned staging buffer (AssociationScan.history_above_msgs / heaps_export_msgs: kgwas_scan_*_msgs), the buffer goes to
# the device in one asynchronous copy, one all_to_all_single moves every rank's messages, and the received words come
# back into a second pinned buffer whose slices the session reads in place (heaps_import / absorb_flat): no host copy
# of the payload on either side.
_STAGE = {}


def _staging(name: str, n_words: int) -> torch.Tensor:
    """Grow-only int64 staging tensor (pinned when the exchange runs over RCCL)."""
    t = _STAGE.get(name)
    if t is None or t.numel() < n_words:
        pin = _dev().type == "cuda"
        t = torch.empty(int(n_words) + int(n_words) // 4 + 1024, dtype=torch.int64, pin_memory=pin)
        _STAGE[name] = t
    return t


def _column_blocks(P: int, world: int) -> np.ndarray:
    """Columns of owner d: blocks[d] .. blocks[d + 1] - 1 (contiguous, so a message is one run of the flat exports)."""
    return np.asarray([(P * d) // world for d in range(world + 1)], np.uint64)


def _pack_msgs_numpy(col0, ncols, counts, kmer, score, row, out):
    """The message writer for scan objects that only offer the flat exports (the pure-Python stand-in of the CPU
    tests): counts[j] entries of column j, flat by column. Returns the lengths; writes if `out` is large enough."""
    counts = np.asarray(counts, np.uint64)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    words = np.asarray([1 + int(n) + 3 * int(counts[int(c):int(c) + int(n)].sum()) for c, n in zip(col0, ncols)], np.uint64)
    if out is None or int(words.sum()) > len(out):
        return words
    o = 0
    for c, n, w in zip(col0, ncols, words):
        c, n, w = int(c), int(n), int(w)
        lo, hi = int(off[c]), int(off[c + n])
        msg = np.concatenate([np.asarray([w - 1], np.int64), counts[c:c + n].view(np.int64), np.asarray(kmer[lo:hi], np.uint64).view(np.int64),
                              np.asarray(score[lo:hi], np.float64).view(np.int64), np.asarray(row[lo:hi], np.uint64).view(np.int64)])
        out[o:o + w] = msg
        o += w
    return words


def _write_msgs(name: str, writer, guess_words: int):
    """Run writer(out) -> message lengths against the staging buffer `name`, growing it once if needed."""
    buf = _staging(name, guess_words)
    words = writer(buf.numpy())
    if int(words.sum()) > buf.numel():
        buf = _staging(name, int(words.sum()))
        words = writer(buf.numpy())
        assert int(words.sum()) <= buf.numel()
    return buf, [int(w) for w in words]


def _history_msgs(scan, thr, col0, ncols, name="send"):
    guess = _STAGE[name].numel() if name in _STAGE else 0
    if hasattr(scan, "history_above_msgs"):
        return _write_msgs(name, lambda out: scan.history_above_msgs(thr, col0, ncols, out), guess)
    flat = scan.history_above(thr)
    return _write_msgs(name, lambda out: _pack_msgs_numpy(col0, ncols, *flat, out), guess)


def _heaps_msgs(scan, col0, ncols, name="send"):
    guess = _STAGE[name].numel() if name in _STAGE else 0
    if hasattr(scan, "heaps_export_msgs"):
        return _write_msgs(name, lambda out: scan.heaps_export_msgs(col0, ncols, out), guess)
    P = scan.n_pheno
    sizes, k, s, r = scan.heaps_export(np.arange(P, dtype=np.uint64))
    return _write_msgs(name, lambda out: _pack_msgs_numpy(col0, ncols, sizes, k, s, r, out), guess)


def _exchange_msgs(send: torch.Tensor, words, name="recv"):
    """send holds this rank's world messages back to back (words[d] words for rank d, each at least the length word).
    Returns the messages received from every rank as views of the staging buffer `name`."""
    world = dist.get_world_size()
    dev = _dev()
    ins = [int(w) for w in words]
    assert len(ins) == world and min(ins) >= 1
    t_in = torch.tensor(ins, dtype=torch.int64, device=dev)
    all_ins = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_ins, t_in)
    rank = dist.get_rank()
    sizes = torch.stack(all_ins).cpu().numpy()
    outs = [int(sizes[src][rank]) for src in range(world)]
    inp = send[:sum(ins)].to(dev, non_blocking=True)
    out = torch.empty(sum(outs), dtype=torch.int64, device=dev)
    dist.all_to_all_single(out, inp, outs, ins)
    recv = _staging(name, sum(outs))
    recv[:sum(outs)].copy_(out)  # device -> pinned (synchronises the stream the exchange ran behind)
    a = recv.numpy()
    off = np.concatenate([[0], np.cumsum(outs)]).astype(np.int64)
    msgs = [a[off[i]:off[i + 1]] for i in range(world)]
    assert all(int(m[0]) == len(m) - 1 for m in msgs)
    return msgs


def _parse_msg(msg, n_cols: int):
    """(counts, kmer, score, row) views of a message of n_cols columns; an empty message reads as no entries."""
    if len(msg) == 1:
        return np.zeros(n_cols, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.float64), np.zeros(0, np.uint64)
    counts = msg[1:1 + n_cols].view(np.uint64)
    n = int(counts.sum())
    body = msg[1 + n_cols:]
    assert len(body) == 3 * n, (len(body), n)
    return counts, body[:n].view(np.uint64), body[n:2 * n].view(np.float64), body[2 * n:].view(np.uint64)


def merge_by_column(scan, dst: int = 0):
    """Column-distributed merge of the shard scans (every rank scanned its contiguous row range; ranks >= 1 with
    record_history). Rank d finishes the columns of block d (_column_blocks): rank 0 ships those heaps' states there
    (layout included), every later rank ships the part of its history that can still matter (score above the larger
    of the earlier shards' final minima), the owner replays shards 1, 2, ... in order - exactly what a single heap
    would have seen - and the final heap states return to rank 0, whose session then holds the global result
    (scan.result(j) after this call). One all_to_all each way; per rank the work is P/G columns x G shards, so
    the merge shrinks with G instead of piling up on rank 0. Returns the total tested-k-mers count.
    `scan` needs: n_pheno, stats(), lowest(), heaps_import(), absorb_flat(), finish() and either the message writers
    history_above_msgs() / heaps_export_msgs() or the flat exports history_above() / heaps_export()."""
    assert dst == 0, "rank 0 holds the heaps of the first shard"
    import os, time
    trace = bool(os.environ.get("KGWAS_TRACE") or os.environ.get("KGWAS_TRACE_MERGE"))
    marks = [("start", time.perf_counter())]
    mark = (lambda name: marks.append((name, time.perf_counter()))) if trace else (lambda name: None)
    rank, world = dist.get_rank(), dist.get_world_size()
    P = scan.n_pheno
    tested = torch.tensor([scan.stats()["rows_tested"]], dtype=torch.int64, device=_dev())
    dist.all_reduce(tested, op=dist.ReduceOp.SUM)
    low, full = scan.lowest()
    lows, fulls = exchange_minima(low, full)
    thr = prefix_thresholds(lows, fulls)
    blocks = _column_blocks(P, world)
    col0, ncols = blocks[:-1].copy(), (blocks[1:] - blocks[:-1]).astype(np.uint64)
    mark("minima")

    # way out: heap states (from rank 0, not to itself) / filtered histories (from ranks >= 1), one message per owner
    if rank == 0:
        out_n = ncols.copy()
        out_n[0] = 0
        send, words = _heaps_msgs(scan, col0, out_n)
    else:
        send, words = _history_msgs(scan, thr[rank], col0, ncols)
    mark("export")
    recv = _exchange_msgs(send, words)
    mark("all_to_all")

    mine = np.arange(int(blocks[rank]), int(blocks[rank + 1]), dtype=np.uint64)
    if len(mine):
        if rank != 0:
            scan.heaps_import(mine, *_parse_msg(recv[0], len(mine)))
            mark("import")
        if world > 1:
            counts = np.zeros((world - 1, P), np.uint64)
            ks, ss, rs = [], [], []
            for g in range(1, world):
                c, k, s, r = _parse_msg(recv[g], len(mine))
                counts[g - 1, int(mine[0]):int(mine[0]) + len(mine)] = c
                ks.append(k); ss.append(s); rs.append(r)
            scan.absorb_flat(counts, ks, ss, rs)  # entries are ordered by column, as counts says
    mark("absorb")

    # way back: final heap states of the owned columns to rank 0
    back_n = np.zeros(world, np.uint64)
    if rank != 0:
        back_n[0] = len(mine)
    send, words = _heaps_msgs(scan, np.full(world, int(blocks[rank]), np.uint64), back_n, name="send_back")
    mark("export_back")
    recv = _exchange_msgs(send, words, name="recv_back")
    mark("all_to_all_back")
    if rank == 0:
        for g in range(1, world):
            n = int(ncols[g])
            if n:
                scan.heaps_import(np.arange(int(col0[g]), int(col0[g]) + n, dtype=np.uint64), *_parse_msg(recv[g], n))
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
    history that can still matter (score above the larger of the earlier shards' final minima; filtered and written
    as one message by the library), rank 0 replays shards 1, 2, ... in order into its own heaps. One exchange, no heap
    state travels; rank 0 does P x N x ln(G) pushes (about a fifth of its own scan's at G = 8), so this is the cheaper
    variant for few ranks, merge_by_column (work per rank P/G columns, but two exchanges of the heap states) for
    many. Returns the total tested-k-mers count. `scan` needs: n_pheno, stats(), lowest(), absorb_flat(), finish()
    and history_above_msgs() or history_above()."""
    assert dst == 0, "rank 0 holds the heaps of the first shard"
    rank, world = dist.get_rank(), dist.get_world_size()
    P = scan.n_pheno
    tested = torch.tensor([scan.stats()["rows_tested"]], dtype=torch.int64, device=_dev())
    dist.all_reduce(tested, op=dist.ReduceOp.SUM)
    low, full = scan.lowest()
    lows, fulls = exchange_minima(low, full)
    thr = prefix_thresholds(lows, fulls)
    col0 = np.zeros(world, np.uint64)
    ncols = np.zeros(world, np.uint64)
    if rank != 0:
        ncols[0] = P
        send, words = _history_msgs(scan, thr[rank], col0, ncols)
    else:
        send, words = _staging("send", world), [1] * world
        send[:world] = 0
    recv = _exchange_msgs(send, words)
    if rank == 0:
        if world > 1:
            counts = np.zeros((world - 1, P), np.uint64)
            ks, ss, rs = [], [], []
            for g in range(1, world):
                c, k, s, r = _parse_msg(recv[g], P)
                counts[g - 1] = c
                ks.append(k); ss.append(s); rs.append(r)
            scan.absorb_flat(counts, ks, ss, rs)
        scan.finish()
    return int(tested.item())


def merge_shards(scan, dst: int = 0):
    """merge_to_root for up to four ranks, merge_by_column beyond (see there). Sessions that keep their columns in select mode
    (kmersgwas_amd/csrc/scan_lazy.cpp: the default without push histories, and every record_history = 2 session) merge to the
    root on logs and pools alone; the by-column merge exports rank 0's heaps in heap-array order, which makes rank 0 replay
    its logs first - so sessions in select mode (`scan.select_mode`, the same on every rank) merge to the root whatever the number
    of ranks: rank 0 then APPENDS the later shards' records to its columns' logs (no heap work; ~7 N records per column at eight
    ranks) and finishes by selection. Who forces merge_by_column creates rank 0's session under KGWAS_FULL_REPLAY=1 (bench.py
    does), so that it replays as it scans."""
    if dist.get_world_size() <= 4 or getattr(scan, "select_mode", False):
        return merge_to_root(scan, dst)
    return merge_by_column(scan, dst)


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
