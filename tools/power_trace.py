"""Board power and shader clock while ONE kernel family runs back to back for a few seconds (verdict r05, item 1b: is the
block-scaled filter's 2.0 GHz a power limit or an issue limit?).

    python tools/power_trace.py [mx|probe|idle] [seconds]

mx:    the headline workload's sparse phase (100 M x 1024 x 101) fed again and again (mx_kernel is 2/3 of the GPU time);
probe: tools/bin/probe_mx6 style bare MFMA stream if built (skipped otherwise);
Samples come from amd-smi / rocm-smi (whatever answers) every ~100 ms on a side thread; the line printed at the end holds the
mean / max socket power, the mean shader clock and the power cap, next to the feed's own kernel times."""
import glob, json, os, subprocess, sys, threading, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "mx"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0


def hwmon_paths():
    out = []
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        p = {}
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input"):
            f = os.path.join(d, name)
            if os.path.exists(f):
                p[name] = f
        if p:
            p["_pci"] = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
            out.append(p)
    return out


def read_int(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None


samples = []
stop = False


def sampler():
    hw = hwmon_paths()
    while not stop:
        t = time.perf_counter()
        s = {"t": t}
        for ci, p in enumerate(hw):
            for k, f in p.items():
                if k.startswith("_"):
                    continue
                v = read_int(f)
                if v is not None:
                    s["%s@%d" % (k, ci)] = v
        if not hw:
            try:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
                j = json.loads(r.stdout)
                c = next(iter(j.values()))
                for k, v in c.items():
                    kl = k.lower()
                    if "power" in kl and "socket" in kl or "average graphics package power" in kl:
                        try:
                            s["smi_power_w"] = float(v)
                        except ValueError:
                            pass
                    if kl.startswith("sclk clock level"):
                        s["smi_sclk"] = v
            except Exception as e:
                s["smi_error"] = str(e)[:80]
        samples.append(s)
        time.sleep(0.1)


import torch
import kmersgwas_amd as kg
from bench import make_phenotypes

th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
n_idle = len(samples)
res = {"mode": mode}
if mode == "mx":
    S, P, M = 1024, 101, 100_000_000
    W = 1 + S // 64
    Y = make_phenotypes(S, P - 1, 7)
    mac = kg.min_count(S, 0.05, 5)
    table = torch.empty(M * W, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    kg.synth_rows_device(table.data_ptr(), 0, M, S, 20240601, stream)
    torch.cuda.synchronize()
    scan = kg.AssociationScan(S, np.arange(S, dtype=np.uint64), Y, 10001, mac, device=0)
    scan.reset(); scan.feed_device(table.data_ptr(), M, 0, stream)
    n_load0 = len(samples)
    t0 = time.perf_counter()
    steps, fk, ak = 0, 0.0, 0.0
    while time.perf_counter() - t0 < secs:
        scan.reset()
        scan.feed_device(table.data_ptr(), M, 0, stream)
        st = scan.stats()
        fk += st["coarse_kernel_ms"]; ak += st["score_kernel_ms"]; steps += 1
    wall = time.perf_counter() - t0
    res.update(steps=steps, wall_s=wall, filter_ms_per_step=fk / steps, kernels_ms_per_step=ak / steps, gpu_duty=ak / steps * steps / (wall * 1e3))
else:
    n_load0 = len(samples)
    time.sleep(secs)
stop = True
th.join()
load = samples[n_load0:]


def col(key, sel):
    v = [s[key] for s in sel if key in s]
    return v


# the host shows every GPU of the node: the one under test is the card whose shader clock is highest under load
n_cards = len(hwmon_paths())
best, best_f = 0, -1.0
for ci in range(n_cards):
    v = col("freq1_input@%d" % ci, load)
    if v and np.mean(v) > best_f:
        best, best_f = ci, float(np.mean(v))
try:
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    for ci, p in enumerate(hwmon_paths()):
        if p["_pci"].lower().startswith(want):
            best = ci
            res["card_by"] = "pci " + p["_pci"]
except Exception as e:
    res["card_by"] = "max clock (%s)" % str(e)[:60]
res["card_index"] = best
res["n_cards"] = n_cards
for name, sel in (("idle", samples[:n_idle]), ("load", load)):
    for key in ("power1_average", "power1_input", "freq1_input", "smi_power_w"):
        v = col("%s@%d" % (key, best), sel) if key != "smi_power_w" else col(key, sel)
        if v:
            scale = 1e-6
            if key == "smi_power_w": scale = 1.0
            res["%s_%s_mean" % (name, key)] = float(np.mean(v)) * scale
            res["%s_%s_min" % (name, key)] = float(np.min(v)) * scale
            res["%s_%s_max" % (name, key)] = float(np.max(v)) * scale
cap = col("power1_cap@%d" % best, samples)
if cap:
    res["power_cap_w"] = cap[0] * 1e-6
res["n_samples_load"] = len(load)
sclk = [s["smi_sclk"] for s in load if "smi_sclk" in s]
if sclk:
    res["smi_sclk_samples"] = sclk[:: max(1, len(sclk) // 8)]
err = [s["smi_error"] for s in samples if "smi_error" in s]
if err:
    res["smi_error"] = err[0]
print(json.dumps(res))
