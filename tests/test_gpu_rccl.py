"""The RCCL ("nccl") branch of the one-process-per-GPU path, executed on the one GPU of the test box.

Every other N > 1 test runs torch.distributed over gloo with CPU tensors; on an 8-GPU node the same code takes the other
branch of kmersgwas_amd/dist.py::_dev(): pinned staging buffers, `.to(cuda, non_blocking=True)`, all_to_all_single /
all_gather / all_reduce on device tensors over RCCL, device -> pinned copies. One rank is legal on one GPU, so these tests
run that branch with world_size 1 - and, so that the exchanges carry real payloads and not only empty messages, route a
second shard's messages (history above the first shard's minima; heap states) through the same `_exchange_msgs` to "rank 0"
itself: the bytes travel host -> pinned -> device -> RCCL all_to_all_single -> device -> pinned -> the library, exactly
the route between two ranks, and the merged heaps must equal a single scan of all rows.
"""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import kmersgwas_amd as kg
    from kmersgwas_amd import dist as kdist
    from helpers import random_table, phenotypes

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and kdist._dev().type == "cuda"
    S, P, N, M = 241, 37, 501, 120_000
    rows = random_table(M, S, seed=5, dup_frac=0.3)
    col = np.arange(S, dtype=np.uint64)
    Y = phenotypes(S, P - 1, seed=9, binary=True)   # binary trait + duplicated patterns: ties everywhere
    mac = kg.min_count(S, 0.05, 5)

    def results(scan):
        return [tuple(a.tobytes() for a in scan.result(j)) for j in range(P)]

    ref = kg.AssociationScan(S, col, Y, N, mac, chunk_rows=8192)
    ref.feed_host(rows)
    ref.finish()
    want, want_tested = results(ref), ref.stats()["rows_tested"]

    # 1. the three merge entry points at world 1: collectives on device tensors, empty messages, results unchanged
    for merge in (kdist.merge_to_root, kdist.merge_by_column, kdist.merge_shards):
        one = kg.AssociationScan(S, col, Y, N, mac, chunk_rows=8192)
        one.feed_host(rows)
        tested = merge(one)
        assert tested == want_tested and results(one) == want, merge.__name__
        one.close()
    assert kdist._STAGE["send"].is_pinned() and kdist._STAGE["recv"].is_pinned()

    # 2. two shards, the second one's messages sent through RCCL to this same rank
    cut = 70_001
    a = kg.AssociationScan(S, col, Y, N, mac, chunk_rows=8192)                       # "rank 0": shard 0
    a.feed_host(rows[:cut], 0)
    b = kg.AssociationScan(S, col, Y, N, mac, chunk_rows=8192, record_history=2)     # "rank 1": shard 1, eviction ring
    b.feed_host(rows[cut:], cut)
    low, full = a.lowest()
    lows, fulls = kdist.exchange_minima(low, full)                                    # all_gather on the device
    assert lows.shape == (1, P) and (lows[0] == low).all() and (fulls[0] == full).all()
    thr = np.where(full, low, -np.inf)
    c0, nc = np.zeros(1, np.uint64), np.asarray([P], np.uint64)
    send, words = kdist._history_msgs(b, thr, c0, nc)                                 # library -> pinned staging
    assert send.is_pinned() and words[0] > 1 + P
    recv = kdist._exchange_msgs(send, words)                                          # pinned -> device -> all_to_all_single -> pinned
    cnt, k, s, r = kdist._parse_msg(recv[0], P)
    assert int(cnt.sum()) > 0
    a.absorb_flat(cnt[None, :], [k], [s], [r])
    a.finish()
    assert results(a) == want, "merged heaps differ from the single scan"
    assert a.stats()["rows_tested"] + b.stats()["rows_tested"] == want_tested

    # 3. heap states (layout included) through the same route: export -> RCCL -> import into a fresh session
    send, words = kdist._heaps_msgs(ref, c0, nc, name="send_back")
    recv = kdist._exchange_msgs(send, words, name="recv_back")
    d = kg.AssociationScan(S, col, Y, N, mac)
    d.heaps_import(np.arange(P, dtype=np.uint64), *kdist._parse_msg(recv[0], P))
    d.finish()
    assert results(d) == want

    # 4. kinship partials: all_reduce of u64 sums viewed as int64 on the device
    kin = kg.Kinship(S, int(np.ceil(S * 0.05)))
    kin.feed_host(rows[:30_000])
    H, n = kin.partials()
    Hs, ns = kdist.allreduce_kinship(H, n)
    assert ns == n and (Hs == H).all()
    dist.barrier()
    dist.destroy_process_group()
    print("rccl-ok")
""")


def _env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                LOCAL_WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_dist_merges_and_exchanges_over_rccl_on_one_rank(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl-ok" in r.stdout


@pytest.mark.parametrize("merge,shape", [("root", "small"), ("column", "small"), ("column", "north_star")])
def test_bench_n_gt_1_path_over_rccl_on_one_rank(merge, shape):
    """bench.py's N > 1 path (init_process_group("nccl", device_id=...), merge inside the timed region, max over ranks,
    per-rank records gathered on device tensors, shard parity check, --check-merge) with one rank on RCCL."""
    env = dict(_env(), KGWAS_BENCH_FORCE_DIST="1", KGWAS_BENCH_MERGE=merge)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "600000",
           "--samples", "241", "--perms", "12", "--topn", "2001", "--check-merge", "--cpu-sample-rows", "300000"]
    if shape == "north_star":  # BASELINE configs[3]'s columns and heap size over RCCL
        cmd[cmd.index("--rows") + 1:cmd.index("--check-merge")] = ["2000000", "--samples", "2048", "--perms", "200", "--topn", "10001"]
        cmd[cmd.index("--cpu-sample-rows") + 1] = "200000"
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["merge_check"] is True and j["parity_check"] is True
    assert [x["rank"] for x in j["ranks"]] == [0] and j["ranks"][0]["merge_ms"] > 0
