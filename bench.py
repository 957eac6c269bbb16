#!/usr/bin/env python3
"""bench.py — association-scan throughput on MI355X.

Metric (BASELINE.json): k-mers x permutations scored per second. A "step" is one full pass of the
hot path (associate_kmers pass 1: MAC filter, score every k-mer against every phenotype column,
top-N heaps with best_associations_heap semantics) over a synthetic table that is already resident
in HBM when the timed region starts. At N=1 the workload is BASELINE.json configs[1]: 100M k-mers x
1024 samples, 1 phenotype + 100 permutations, top 10001 per column. With N>1 every rank scans its
own 100M-row shard (weak scaling) and rank 0 merges the ranks' heap histories over RCCL.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
I8_MFMA_PEAK_TOPS = 5000.0    # dense int8 (= fp8 rate, 2x bf16); measured ceiling 3944-4404 TOPS (same guide)
HBM_PEAK_GBPS = 8000.0


def make_phenotypes(S, n_perm, seed):
    rng = np.random.default_rng(seed)
    y0 = rng.standard_normal(S).astype(np.float32)
    cols = [y0]
    for p in range(n_perm):
        cols.append(np.random.default_rng(seed + 1 + p).permutation(y0))
    return np.ascontiguousarray(np.stack(cols).astype(np.float32))


def usable_cpus():
    """CPUs this process may really use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        pass
    return n


def cpu_baseline(S, Y, mac, topn, seed, sample_rows, threads):
    """Oracle (CPU restatement of the reference algorithm: per-bit squeeze loader + SSE-order
    scorer + std::priority_queue, one thread per phenotype column) on a bounded sample."""
    import kmersgwas_amd as kg
    from oracle import binding as ob
    rows = kg.synth_rows_host(0, sample_rows, S, seed)
    t0 = time.perf_counter()
    res = ob.associate(rows, S, np.arange(S, dtype=np.uint64), Y, topn, mac, batch_size=10_000_000, threads=threads)
    dt = time.perf_counter() - t0
    return dict(value=sample_rows * Y.shape[0] / dt, unit="kmer*phenotype/s", cores=threads, kind="port",
                sample="%d rows x %d samples x %d columns of the same synthetic table; oracle/oracle.cpp "
                       "(load %.2fs + score %.2fs)" % (sample_rows, S, Y.shape[0], res["t_load"], res["t_score"]),
                seconds=dt)


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this container's CPU controller, (0, 0) if unreadable."""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except Exception:
        return 0, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=100_000_000, help="k-mer rows per GPU")
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--perms", type=int, default=100)
    ap.add_argument("--topn", type=int, default=10001)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--cpu-sample-rows", type=int, default=6_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import kmersgwas_amd as kg
    from kmersgwas_amd import dist as kdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        if ndev < int(os.environ.get("LOCAL_WORLD_SIZE", str(world))):
            # several ranks on one GPU (only done to exercise the N>1 path on a 1-GPU box): the library would pin
            # every rank's replay workers to the same cores of that GPU's NUMA share
            os.environ.setdefault("KGWAS_PIN_THREADS", "0")
        torch.cuda.set_device(local_rank % ndev)
        # KGWAS_DIST_BACKEND=gloo lets several ranks share one GPU (used to exercise the N>1 path on a 1-GPU box)
        backend = os.environ.get("KGWAS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()

    S, P, M = args.samples, args.perms + 1, args.rows
    W = 1 + (S + 63) // 64
    seed_table, seed_y = 20240601, 7
    Y = make_phenotypes(S, args.perms, seed_y)
    mac = kg.min_count(S, 0.05, 5)
    col = np.arange(S, dtype=np.uint64)

    # Table shard resident in HBM before timing (generated on the device; never touches PCIe).
    table = torch.empty(M * W, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    first_row = rank * M
    kg.synth_rows_device(table.data_ptr(), first_row, M, S, seed_table, stream)
    torch.cuda.synchronize()

    # One scan session per process, reused across steps (kgwas_scan_reset empties heaps and statistics
    # but keeps device / pinned buffers): a step is one full pass of the hot path, not buffer set-up.
    # Host replay pool: this rank's share of the CPUs the job may use (one rank per GPU on one node).
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    host_threads = max(1, usable_cpus() // max(local_world, 1))
    session = kg.AssociationScan(S, col, Y, args.topn, mac, device=dev, kernel=args.kernel,
                                 chunk_rows=args.chunk_rows, record_history=(2 if (world > 1 and rank > 0) else 0), host_threads=host_threads)

    merge_ms = []

    def one_step():
        scan = session
        scan.reset()
        scan.feed_device(table.data_ptr(), M, first_row, stream)
        if world == 1:
            scan.finish()  # with several ranks the merge finishes rank 0's session once, at its end
        st = scan.stats()
        heaps = None
        if world > 1:
            if os.environ.get("KGWAS_BENCH_MERGE_DIAG"):  # diagnostics: separate "waiting for the slowest rank" from the merge
                dist.barrier()
            tm = time.perf_counter()
            merge = {"root": kdist.merge_to_root, "column": kdist.merge_by_column}.get(os.environ.get("KGWAS_BENCH_MERGE", ""), kdist.merge_shards)
            tested = merge(scan)  # rank 0's session now holds the global heaps
            merge_ms.append((time.perf_counter() - tm) * 1e3)
        else:
            tested = st["rows_tested"]
        return scan, st, heaps, tested

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        scan, st, heaps, tested = one_step()
    sync()
    thr0 = cgroup_throttle()
    t0 = time.perf_counter()
    stats = []
    step_ms = []
    last = None
    for _ in range(args.steps):
        ts = time.perf_counter()
        last, st, heaps, tested = one_step()
        stats.append(st)
        step_ms.append((time.perf_counter() - ts) * 1e3)
    sync()
    dt = time.perf_counter() - t0
    thr1 = cgroup_throttle()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=kdist._dev())
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        total_rows = M * world
        value = total_rows * P / (ms_per_step / 1e3)
        k_ms = sum(s["score_kernel_ms"] for s in stats)
        k_launch = sum(s["score_launches"] for s in stats)
        rows_scored = M * args.steps  # this rank
        flop_per_row = 2.0 * S * P    # 2*S flop per (k-mer, phenotype)
        avg_ms = k_ms / max(k_launch, 1)
        achieved_tflops = flop_per_row * rows_scored / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ku = stats[-1]["kernel_used"]
        kernel_name = {1: "score_valu_kernel", 2: "score_mfma_kernel", 3: "coarse_kernel"}[ku]
        peak, peak_unit, dtype = F32_MFMA_PEAK_TFLOPS, "TFLOP/s", "f32"
        executed = None
        hbm_bound = False
        if ku == 3:
            # dominant kernel = the int8 coarse filter; its own launches are timed separately from the exact
            # re-scoring of the survivors. `achieved` stays ALGORITHMIC (2*S flop per k-mer x column); the
            # kernel executes 2 int8 slices per column and pads columns/samples to its tiles (`executed`).
            k_ms = sum(s["coarse_kernel_ms"] for s in stats)
            k_launch = sum(s["coarse_launches"] for s in stats)
            rows_scored = sum(s["rows_fed"] for s in stats) - 16384 * args.steps  # the dense chunk uses the exact kernel
            avg_ms = k_ms / max(k_launch, 1)
            achieved_tflops = flop_per_row * rows_scored / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
            peak, peak_unit, dtype = I8_MFMA_PEAK_TOPS, "TOP/s (int8 MFMA dense)", "i8 filter + f32/f64 exact re-score"
            Wm = 2 * ((S + 127) // 128)
            kgroups = (Wm + 7) // 8
            # executed int8 work: per operand set, padded samples x padded operand columns x rows filtered with it
            ex_ops = 0.0
            for mi in range(2):
                T, lgroups = stats[-1]["coarse_mode_tiles"][mi], stats[-1]["coarse_mode_lgroups"][mi]
                ex_ops += 2.0 * (kgroups * 512) * (lgroups * T * 16) * sum(st_["coarse_mode_rows"][mi] for st_ in stats)
            rows_scored = sum(sum(st_["coarse_mode_rows"]) for st_ in stats)
            achieved_tflops = flop_per_row * rows_scored / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
            executed = ex_ops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
            # roofline side: 2*S*P op per 8*W bytes against the balance point of the int8 pipe and HBM
            if flop_per_row / (8.0 * W) < I8_MFMA_PEAK_TOPS * 1e12 / (HBM_PEAK_GBPS * 1e9):
                hbm_bound = True
                peak, peak_unit = HBM_PEAK_GBPS, "GB/s (HBM3E)"
                achieved_tflops = rows_scored * 8.0 * W / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0  # algorithmic bytes / s
        # sanity on the final result of the last step: ascending pops, full heaps
        k, sc, r = last.result(0)
        assert (np.diff(sc) >= 0).all() and len(k) == min(args.topn, tested)
        # HBM traffic per launch cannot be read from inside the process; it comes from the committed
        # rocprofv3 PMC passes of this same workload (FETCH_SIZE x2 on gfx950 + WRITE_SIZE).
        traffic, traffic_src = None, None
        pmc_name = "r01b_coarse_pmc_hbm_traffic.json" if ku == 3 else "r01a_exactmfma_pmc_hbm_traffic.json"
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if S == 1024 and P == 101 and ku in (2, 3) and os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                # PMC bytes per row of the steady launches -> GB per average launch of this run
                traffic = j["traffic_bytes_per_row"] * (rows_scored / max(k_launch, 1)) / 1e9
                traffic_src = "profiles/" + pmc_name
            except Exception:
                pass
        out = {
            "metric": "k-mers x permutations scored/sec", "value": value, "unit": "kmer*phenotype/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "synthetic %dM k-mers x %d samples per GPU, 1 phenotype + %d permutations, "
                                   "top-%d per column, maf 0.05 / mac 5 (BASELINE.json configs[1])"
                                   % (M // 1_000_000, S, args.perms, args.topn),
                       "rows_per_gpu": M, "samples": S, "phenotype_columns": P, "topn": args.topn,
                       "min_count": int(mac), "sharding": "rows, contiguous per rank" if world > 1 else "none"},
            "rows_per_s": total_rows / (ms_per_step / 1e3),
            "hbm_read_GBps_algorithmic": total_rows * 8.0 * W / (ms_per_step / 1e3) / 1e9,
            "rows_tested": int(tested),
            "roofline": {"bound": "hbm" if hbm_bound else ("mfma" if ku in (2, 3) else "valu"), "kernel": kernel_name,
                         "achieved": achieved_tflops, "peak": peak, "unit": peak_unit,
                         "frac": achieved_tflops / peak, "executed_TOPs": executed,
                         "coarse_sets": ([{"int8_slices": mi + 1, "tiles_per_lds_group": stats[-1]["coarse_mode_tiles"][mi],
                                            "lds_groups": stats[-1]["coarse_mode_lgroups"][mi],
                                            "launches_per_step": sum(st_["coarse_mode_launches"][mi] for st_ in stats) / args.steps,
                                            "rows_per_step": sum(st_["coarse_mode_rows"][mi] for st_ in stats) // args.steps,
                                            "ms_per_step": sum(st_["coarse_mode_ms"][mi] for st_ in stats) / args.steps}
                                           for mi in range(2) if stats[-1]["coarse_mode_tiles"][mi]] if ku == 3 else None),
                         "executed_frac": (executed / peak) if executed else None, "traffic": traffic,
                         "traffic_unit": "GB per average launch (HBM-side, PMC)", "traffic_source": traffic_src,
                         "algorithmic_GB_per_launch": rows_scored / max(k_launch, 1) * 8.0 * W / 1e9,
                         "launches": k_launch, "avg_launch_ms": avg_ms,
                         "kernel_ms_per_step": k_ms / args.steps,
                         "all_scoring_kernels_ms_per_step": sum(s["score_kernel_ms"] for s in stats) / args.steps,
                         "hbm_frac_of_8TBps": (rows_scored * 8.0 * W / (k_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS
                         if k_ms > 0 else 0.0},
            "host": {"replay_ms_per_step": sum(s["replay_ms"] for s in stats) / args.steps,
                     "candidates_per_step": sum(s["candidates"] for s in stats) // args.steps,
                     "heap_pushes_per_step": sum(s["heap_pushes"] for s in stats) // args.steps,
                     "chunks_per_step": sum(s["chunks"] for s in stats) // args.steps,
                     "gpu_wait_ms_per_step": sum(s["gpu_wait_ms"] for s in stats) / args.steps,
                     "dense_phase_ms_per_step": sum(s["dense_ms"] for s in stats) / args.steps,
                     "cores": usable_cpus(), "replay_threads_per_rank": host_threads, "logical_cpus": os.cpu_count(),
                     "step_ms": [round(x, 2) for x in step_ms],
                     # cross-shard merge on rank 0's clock (includes waiting for the slowest rank's scan)
                     "merge_ms": [round(x, 2) for x in merge_ms[args.warmup:]],
                     # CFS bandwidth throttling of this container during the timed region (cpu.stat deltas)
                     "cgroup_nr_throttled": thr1[0] - thr0[0], "cgroup_throttled_ms": (thr1[1] - thr0[1]) / 1e3},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, Y, mac, args.topn, seed_table, args.cpu_sample_rows,
                                               threads=min(usable_cpus(), P))
        print(json.dumps(out))
    if last is not None:
        last.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
