"""bench.py's headline with 2 and 16 replay threads, logs by reference (default) and by copy (KGWAS_LOG_BY_REF=0)."""
import subprocess, os, json, sys
args = sys.argv[1:]
for by_ref in ("1", "0"):
    for thr, steps in (("2", "5"), ("16", "10")):
        env = dict(os.environ, KGWAS_HOST_THREADS=thr, KGWAS_LOG_BY_REF=by_ref)
        o = subprocess.run(["python", "bench.py", "--steps", steps, "--warmup", "2", "--no-cpu-baseline", "--no-subrecords"] + args, env=env, capture_output=True, text=True)
        try:
            d = json.loads(o.stdout.strip().splitlines()[-1])
        except Exception:
            print(o.stdout[-500:], o.stderr[-1500:])
            continue
        h = d["host"]
        print("by_ref %s threads %s: %.2f ms/step, replay cpu %.1f ms, busiest %.1f, tail %.2f, parity %s" % (by_ref, thr, d["ms_per_step"], h["replay_cpu_ms_per_step"], h["replay_ms_per_step"], h["replay_tail_ms_per_step"], d.get("parity_check")), flush=True)
