"""Randomised parity runs of the scan against the oracle (diagnostics; the fixed cases live in tests/): random accession
counts, column counts, heap sizes, chunk sizes, feeds, column subsets / orders, tie densities, phenotype kinds, filter forms.
   python tools/fuzz_parity.py [seconds=240] [seed=1] [big]     (needs a GPU)
"big": tables of 0.3-3 M rows (many chunks in flight, record-ring wrap, group splits, popping ahead) instead of 200-60 000."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kmersgwas_amd as kg
from oracle import binding as ob, oracle_np as onp
from helpers import random_table, phenotypes

import faulthandler, threading
_progress = [time.time(), "start"]
def _watchdog(limit=60.0):
    # a scan that makes no progress for `limit` seconds: say where every thread is, then leave (a hang must not eat the call)
    while True:
        time.sleep(5.0)
        if time.time() - _progress[0] > limit:
            sys.stdout.write("HANG: no progress for %.0f s in: %s\n" % (time.time() - _progress[0], _progress[1])); sys.stdout.flush()
            faulthandler.dump_traceback(file=sys.stdout, all_threads=True); sys.stdout.flush()
            try:
                for t in sorted(os.listdir("/proc/self/task"), key=int):
                    def rd(f):
                        try: return open("/proc/self/task/%s/%s" % (t, f)).read().strip()
                        except Exception as e: return "?"
                    st = rd("stat").split(") ")[-1].split()
                    sys.stdout.write("tid %s comm %-16s state %s cpu_ticks %s+%s wchan %-28s syscall %s\n" % (t, rd("comm"), st[0], st[11] if len(st) > 12 else "?", st[12] if len(st) > 12 else "?", rd("wchan"), rd("syscall")[:40]))
            except Exception as e:
                sys.stdout.write("proc scan failed: %r\n" % (e,))
            sys.stdout.flush()
            os._exit(3)
threading.Thread(target=_watchdog, daemon=True).start()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
t_end = time.time() + budget
n_ok = 0
while time.time() < t_end:
    S_f = int(rng.choice([5, 33, 64, 100, 241, 257, 511, 512, 700, 1024, 1135, 1500, 2048, 2600]))
    reorder = rng.random() < 0.3
    S = S_f if not reorder else int(rng.integers(max(2, S_f // 3), S_f + 1))
    P = int(rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 47, 64, 101, 130]))
    n_rows = int(rng.integers(200, 60_000))
    topn = int(rng.choice([1, 2, 7, 64, 301, 1000, 5000]))
    chunk = int(rng.choice([0, 128, 1024, 4096, 8192, 65536]))
    if BIG:
        S_f = int(rng.choice([241, 512, 1024, 1135, 2048]))
        S = S_f if not reorder else int(rng.integers(S_f // 2, S_f + 1))
        P = int(rng.choice([1, 3, 4, 16, 40, 101]))
        n_rows = int(rng.integers(300_000, 3_000_000) * (1.0 if P <= 16 else 0.4))
        topn = int(rng.choice([301, 5000, 10001]))
        chunk = int(rng.choice([0, 0, 65536, 262144]))
    dup = float(rng.choice([0.0, 0.0, 0.2, 0.6]))
    kind = str(rng.choice(["normal", "binary", "shifted", "heavy", "tiny", "ints", "subnormal", "near_overflow", "onehot"]))
    kernel = int(rng.choice([kg.KERNEL_AUTO, kg.KERNEL_AUTO, kg.KERNEL_COARSE, kg.KERNEL_MFMA, kg.KERNEL_VALU]))
    env = {"KGWAS_COARSE_MX": str(rng.choice(["", "0", "1"])),
           "KGWAS_COARSE_SLICES": str(rng.choice(["", "", "1", "2"])),
           "KGWAS_MXS": str(rng.choice(["", "", "0", "2", "3", "3"])), "KGWAS_MXS_FORM": str(rng.choice(["", "", "1", "2"])),
           "KGWAS_FULL_REPLAY": str(rng.choice(["", "", "", "1"])), "KGWAS_LOG_BY_REF": str(rng.choice(["", "", "0"])),
           "KGWAS_RING_BYTES": str(rng.choice(["", "", "1", "3000000"])), "KGWAS_DSEL_REG": str(rng.choice(["", "", "0"])),
           "KGWAS_TAIL_KERNEL": str(rng.choice(["", "", "0"])), "KGWAS_TIE_CHECKS": str(rng.choice(["", "", "0", "1"])),
           "KGWAS_FINISH_THREADS": str(rng.choice(["", "", "7"]))}
    for k, v in env.items():
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    rows = random_table(n_rows, S_f, seed=int(rng.integers(1 << 30)), dup_frac=dup)
    col = rng.permutation(S_f)[:S].astype(np.uint64) if reorder else np.arange(S, dtype=np.uint64)
    Y = phenotypes(S, P - 1, seed=int(rng.integers(1 << 30)), binary=(kind == "binary"))
    if kind == "shifted": Y = (Y + np.float32(rng.choice([100.0, -7.5, 1e4]))).astype(np.float32)
    if kind == "heavy": Y = Y.copy(); Y[:, 0] += np.float32(50.0)
    if kind == "tiny": Y = (Y * np.float32(1e-30)).astype(np.float32)
    if kind == "ints": Y = np.round(Y * 20).astype(np.float32)
    if kind == "subnormal": Y = (Y * np.float32(rng.choice([1e-38, 1e-41, 1e-44]))).astype(np.float32)
    if kind == "near_overflow":  # sum |y| of a column between 0.3 and 3 x FLT_MAX: on both sides of the filters' gate
        a = np.abs(Y.astype(np.float64)).sum(axis=1).max()
        Y = (Y.astype(np.float64) * (float(rng.uniform(0.3, 3.0)) * 3.4e38 / max(a, 1e-30))).astype(np.float32)
        Y[~np.isfinite(Y)] = np.float32(3e38)
    if kind == "onehot":
        Y = Y.copy(); Y[0] = 0; Y[0, int(rng.integers(S))] = np.float32(2.5)
    Y = np.ascontiguousarray(Y, np.float32)
    mac = onp.min_count(S, 0.05, 5) if S >= 100 else 1
    desc = dict(S_f=S_f, S=S, P=P, n_rows=n_rows, topn=topn, chunk=chunk, dup=dup, kind=kind, kernel=kernel, env=env, reorder=bool(reorder))
    if os.environ.get("KGWAS_FUZZ_VERBOSE"):  # name the scan BEFORE it runs: a hang then shows its configuration
        print("scan %d:" % n_ok, desc, flush=True)
    try:
        if kernel == kg.KERNEL_MFMA and S > 2600: kernel = kg.KERNEL_AUTO
        _progress[:] = [time.time(), "oracle %r" % (desc,)]
        exp = ob.associate(rows, S_f, col, Y, topn, mac, batch_size=int(rng.integers(100, 20000)), threads=16 if BIG else 4)
        _progress[:] = [time.time(), "create %r" % (desc,)]
        scan = kg.AssociationScan(S_f, col, Y, topn, mac, kernel=kernel, chunk_rows=chunk, host_threads=int(rng.choice([0, 1, 3, 8])))
        cuts = sorted(set([0, n_rows] + [int(x) for x in rng.integers(0, n_rows + 1, size=int(rng.integers(0, 3)))]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi == n_rows and rng.random() < 0.5: scan.expect_finish()
            _progress[:] = [time.time(), "feed_host [%d, %d) %r" % (lo, hi, desc)]
            scan.feed_host(rows[lo:hi], lo)
        _progress[:] = [time.time(), "finish %r" % (desc,)]
        scan.finish()
        st = scan.stats()
        assert st["columns_selected"] > 0 or st["heap_pushes"] == exp["pushes"], ("pushes", st["heap_pushes"], exp["pushes"])
        assert st["rows_tested"] == exp["tested"], ("tested", st["rows_tested"], exp["tested"])
        for j in range(P):
            k, s, r = scan.result(j)
            o = exp["per_pheno"][j]
            assert len(k) == len(o["kmer"]) and (k == o["kmer"]).all() and (r == o["file_row"]).all() and s.tobytes() == o["score"].tobytes(), ("column", j)
        _progress[:] = [time.time(), "close %r" % (desc,)]
        scan.close()
        _progress[:] = [time.time(), "between scans"]
        n_ok += 1
    except kg.KgwasError as e:
        if "coarse filter" in str(e) or "MFMA scorer" in str(e):  # a forced kernel that does not apply to this shape
            continue
        print("ERROR", desc, e); sys.exit(1)
    except AssertionError as e:
        print("MISMATCH", desc, e); sys.exit(1)
print("fuzz: %d random scans equal the oracle's" % n_ok)
