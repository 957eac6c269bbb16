"""world_size-2, -5 and -8 `gloo` tests of the multi-GPU plumbing on CPU (kmersgwas_amd/dist.py).

The product's scoring needs a GPU, so each rank's shard-local heap-push history and kinship partials
are produced here by the oracle (as the checker / stand-in data source); what is under test is the
N>1 path itself: shard ranges, the history gather over torch.distributed, the in-order replay on rank 0
(kgwas_merge_shards through the C ABI) and the integer all-reduce of kinship partials."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import kmersgwas_amd as kg
    from kmersgwas_amd import dist as kdist
    from oracle import binding as ob
    from oracle import oracle_np as onp
    from helpers import random_table, phenotypes
    from test_host import _python_history

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    S_f, S, P, N, M = 90, 80, 3, 40, 2500
    rows = random_table(M, S_f, seed=41, dup_frac=0.4)          # same table on every rank
    col = np.random.default_rng(2).permutation(S_f)[:S].astype(np.uint64)
    Y = phenotypes(S, P - 1, seed=6, binary=True)
    mac = onp.min_count(S, 0.05, 5)
    lo, hi = kdist.shard_range(M, rank, world)
    assert (lo, hi) == ((M * rank) // world, (M * (rank + 1)) // world)
    dense, kept = ob.scores_dense(rows[lo:hi], S_f, col, Y, mac)  # stand-in for this rank's GPU scan
    idx = np.nonzero(kept)[0]
    hist = [_python_history(rows[lo:hi][idx, 0], dense[j][idx], (idx + lo).astype(np.uint64), N) for j in range(P)]
    shards = kdist.gather_histories(hist, dst=0)
    if rank == 0:
        assert len(shards) == world
        heaps = kg.merge_shards(N, shards, threads=2)
        exp = ob.associate(rows, S_f, col, Y, N, mac)
        for j in range(P):
            k, s, r = heaps[j].pop_all()
            o = exp["per_pheno"][j]
            assert (k == o["kmer"]).all() and (r == o["file_row"]).all() and s.tobytes() == o["score"].tobytes()
    else:
        assert shards is None
    # The cheap exchange bench.py uses: final minima all-gathered, later shards pre-filtered by
    # score > max(minima of earlier full heaps); shard 0 is sent in full here only because this CPU test
    # has no scan session to absorb into (on GPUs rank 0 keeps its heaps and absorbs shards 1..).
    py = [onp.BestHeap(N) for _ in range(P)]
    for j in range(P):
        for kk_, ss_, rr_ in zip(*hist[j]):
            py[j].add(int(kk_), float(ss_), int(rr_))
    low = np.asarray([h.lowest for h in py]); full = np.asarray([len(h.q) >= N for h in py])
    lows, fulls = kdist.exchange_minima(low, full)
    thr = kdist.prefix_thresholds(lows, fulls)
    assert (thr[0] == -np.inf).all()
    filt = hist if rank == 0 else kdist.filter_history(hist, thr[rank])
    if rank > 0:
        assert sum(len(h[0]) for h in filt) < sum(len(h[0]) for h in hist)
    shards2 = kdist.gather_histories(filt, dst=0)
    if rank == 0:
        heaps2 = kg.merge_shards(N, shards2, threads=2)
        for j in range(P):
            k, s, r = heaps2[j].pop_all()
            o = exp["per_pheno"][j]
            assert (k == o["kmer"]).all() and (r == o["file_row"]).all() and s.tobytes() == o["score"].tobytes()
    # The column-distributed merge bench.py uses (kdist.merge_by_column): heap states travel to the column's owner
    # rank in heap-array order, filtered histories follow, final states return to rank 0. The scan session is
    # stood in for by the pure-Python heap restatement (layout-exact: its array IS libstdc++'s).
    class PyScan:
        def __init__(self, hist_):
            self.n_pheno = P
            self.hist = hist_
            self.heaps = [onp.BestHeap(N) for _ in range(P)]
            for j in range(P):
                for kk_, ss_, rr_ in zip(*hist_[j]):
                    self.heaps[j].add(int(kk_), float(ss_), int(rr_))
        def stats(self):
            return {"rows_tested": int(kept.sum())}
        def lowest(self):
            return (np.asarray([h.lowest for h in self.heaps]), np.asarray([len(h.q) >= N for h in self.heaps]))
        def history_above(self, t):
            f = kdist.filter_history(self.hist, t)
            cnt = np.asarray([len(x[0]) for x in f], np.uint64)
            cat = lambda i, dt: np.concatenate([np.asarray(x[i], dt) for x in f]) if cnt.sum() else np.zeros(0, dt)
            return cnt, cat(0, np.uint64), cat(1, np.float64), cat(2, np.uint64)
        def heaps_export(self, cols):
            v = [self.heaps[int(j)].q.v for j in cols]
            flat = [e for x in v for e in x]
            return (np.asarray([len(x) for x in v], np.uint64), np.asarray([e[0] for e in flat], np.uint64),
                    np.asarray([e[1] for e in flat], np.float64), np.asarray([e[2] for e in flat], np.uint64))
        def heaps_import(self, cols, sizes, k, s, r):
            o = 0
            for j, n in zip(cols, sizes):
                h = self.heaps[int(j)]
                h.q.v = [(int(k[o + i]), float(s[o + i]), int(r[o + i])) for i in range(int(n))]
                h.lowest = h.q.top()[1] if int(n) else 0.0
                o += int(n)
        def absorb_flat(self, counts, ks, ss, rs):
            for g in range(counts.shape[0]):
                o = 0
                for j in range(P):
                    for i in range(int(counts[g, j])):
                        self.heaps[j].add(int(ks[g][o + i]), float(ss[g][o + i]), int(rs[g][o + i]))
                    o += int(counts[g, j])
        def finish(self):
            pass
    by_column_calls = []
    _by_column = kdist.merge_by_column
    def _spy(scan, dst=0):
        by_column_calls.append(1)
        return _by_column(scan, dst)
    kdist.merge_by_column = _spy  # (merge_shards looks the name up when it is called)
    for merge in (kdist.merge_by_column, kdist.merge_to_root, kdist.merge_shards):
        ps = PyScan(hist)
        tested = merge(ps)
        if rank == 0:
            assert tested == exp["tested"], (tested, exp["tested"])
            for j in range(P):
                pops = ps.heaps[j].pop_all()
                o = exp["per_pheno"][j]
                assert [e[0] for e in pops] == [int(x) for x in o["kmer"]], "column %%d" %% j
                assert [e[2] for e in pops] == [int(x) for x in o["file_row"]]
                assert np.asarray([e[1] for e in pops]).tobytes() == o["score"].tobytes()
    # merge_shards picks by itself: to the root up to four ranks, by column beyond
    assert len(by_column_calls) == (2 if world > 4 else 1), (world, by_column_calls)
    # kinship partials: integer Hamming sums + used-row counts all-reduce to the single-process answer
    mc = int(np.ceil(S_f * 0.05))
    g = onp.unpack_bits(rows[lo:hi], np.arange(S_f, dtype=np.uint64)).astype(np.int64)
    n1 = g.sum(axis=1)
    g = g[(n1 >= mc) & (n1 <= S_f - mc)]
    H = (g[:, :, None] ^ g[:, None, :]).sum(axis=0).astype(np.uint64)
    Hs, n = kdist.allreduce_kinship(H, len(g))
    K = kg.kinship_from_partials(Hs, n)
    Ko, no = ob.kinship(rows, S_f, mc)
    assert n == no and (K == Ko).all()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank%%d-ok\\n" %% rank); sys.stdout.flush()
""")


import pytest


# 5 and 8 ranks > 3 columns: ranks that own no column take part with empty messages; 8 (the node's GPU count): merge_shards
# takes merge_by_column by itself
@pytest.mark.parametrize("world", [2, 5, 8])
def test_gloo_merge_and_kinship_allreduce(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("-ok") == world, r.stdout
