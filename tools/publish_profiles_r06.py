"""Copy the rocprofv3 summaries tools/collect_profiles_r06.sh left under gpurun_out/prof_r06/ into profiles/ (the tracked, judged
copies): kernel statistics (bench workload, one-column scan, kinship, the 2048 x 201 shape), PMC passes (HBM traffic and SQ counters of
mx_kernel), the power probe and the bench line. Usage: python tools/publish_profiles_r06.py"""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, ".")
src = "gpurun_out/prof_r06"
os.makedirs("profiles", exist_ok=True)
from bench import kernel_source_sha16


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)


def strip_stats(path, out, keep=25):
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:keep]:
            n = r["Name"]
            if len(n) > 160:
                n = n[:60] + " ... " + n[-60:]
            w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])


def launches(pass_dir, kname):
    """The kernel's launches of one profiler pass in dispatch order: [{id, dur_ns, counters{}}]."""
    by_id = collections.OrderedDict()
    for r in csv.DictReader(open(one(pass_dir + "/*/*_kernel_trace.csv"))):
        if kname in r["Kernel_Name"]:
            by_id[int(r["Dispatch_Id"])] = {"id": int(r["Dispatch_Id"]), "dur_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}}
    for r in csv.DictReader(open(one(pass_dir + "/*/*_counter_collection.csv"))):
        if kname in r["Kernel_Name"] and int(r["Dispatch_Id"]) in by_id:
            by_id[int(r["Dispatch_Id"])]["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return [by_id[k] for k in sorted(by_id)]


def steady(ls):
    """Persistent blocks: every launch of 131 072 rows and more has the same grid; the steady launches (8 388 608 rows) are the
    longest ones - within 10 % of the longest."""
    top = max(x["dur_ns"] for x in ls)
    return [x for x in ls if x["dur_ns"] >= 0.9 * top]


NOTE = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/collect_profiles_r06.sh); counter values are KiB; on "
        "gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so HBM-side read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM "
        "section); WRITE_SIZE is taken as is. Steady launches = the longest launches of the pass (within 10 %): with persistent blocks the "
        "grid is the same for every chunk of 131 072 rows and more.")

# ---- mx_kernel at 1024 x 101: traffic
KN, ROWS = "mx_kernel", 8388608
lf, lw = steady(launches("pmc_fetch", KN)), steady(launches("pmc_write", KN))
fv = sum(x["c"]["FETCH_SIZE"] for x in lf) / len(lf)
wv = sum(x["c"]["WRITE_SIZE"] for x in lw) / len(lw)
t = 2.0 * fv * 1024.0 + wv * 1024.0
json.dump({"kernel": KN, "kernel_source_sha16": kernel_source_sha16("score_mx.hip"),
           "fetch": {"counter": "FETCH_SIZE", "launches_averaged": len(lf), "value_KiB_per_launch": fv},
           "write": {"counter": "WRITE_SIZE", "launches_averaged": len(lw), "value_KiB_per_launch": wv},
           "rows_per_launch": ROWS, "algorithmic_bytes_per_launch": ROWS * 136, "traffic_bytes_per_launch": t,
           "traffic_bytes_per_row": t / ROWS, "traffic_over_algorithmic": t / (ROWS * 136), "note": NOTE},
          open("profiles/r06_mx_pmc_hbm_traffic.json", "w"), indent=1)
print("mx_kernel 1024 x 101: traffic / algorithmic = %.3f (%d + %d launches)" % (t / (ROWS * 136), len(lf), len(lw)))

strip_stats(one("stats/*/*_kernel_stats.csv"), "profiles/r06_bench_kernel_stats.csv")
strip_stats(one("p1_stats/*/*_kernel_stats.csv"), "profiles/r06_p1scan_kernel_stats.csv")
strip_stats(one("kin_stats/*/*_kernel_stats.csv"), "profiles/r06_kinship_kernel_stats.csv")
strip_stats(one("c3_stats/*/*_kernel_stats.csv"), "profiles/r06_config4_kernel_stats.csv")
line = [l for l in open(os.path.join(src, "bench_line.json")) if l.startswith("{")][-1]
open("profiles/r06_bench_line.json", "w").write(line)
for name in ("probe_mx_power.txt", "power_trace_mx.json", "probe_events.txt", "probe_d2h.txt", "chunk_timeline.txt", "steps_final.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), "profiles/r06_" + name)

# ---- the resident plan at 2048 x 201 (five LDS groups x 768 threads per row block of 4096 rows: not persistent)
lf, lw = launches("pmc_c3r_fetch", "mx_kernel"), launches("pmc_c3r_write", "mx_kernel")
lf, lw = steady(lf), steady(lw)
fv = sum(x["c"]["FETCH_SIZE"] for x in lf) / len(lf)
wv = sum(x["c"]["WRITE_SIZE"] for x in lw) / len(lw)
# rows of the steady launches: the chunk cap at this shape (32-bit byte offsets: < 4 GiB of rows per chunk)
rows_c3 = ((1 << 32) - (1 << 20)) // 264 // 128 * 128
rows_c3 = min(rows_c3, 8388608)
tr = 2.0 * fv * 1024.0 + wv * 1024.0
json.dump({"kernel": "mx_kernel", "kernel_source_sha16": kernel_source_sha16("score_mx.hip"),
           "shape": "2048 samples x 201 columns: the default resident plan, five LDS groups of three column tiles per row block",
           "fetch": {"counter": "FETCH_SIZE", "launches_averaged": len(lf), "value_KiB_per_launch": fv},
           "write": {"counter": "WRITE_SIZE", "launches_averaged": len(lw), "value_KiB_per_launch": wv},
           "rows_per_launch_upper_bound": rows_c3, "traffic_bytes_per_launch": tr,
           "traffic_over_algorithmic_lower_bound": tr / (rows_c3 * 264), "note": NOTE},
          open("profiles/r06_mx_pmc_hbm_traffic_2048x201.json", "w"), indent=1)
print("mx_kernel (resident, 5 LDS groups) 2048 x 201: traffic / algorithmic >= %.3f" % (tr / (rows_c3 * 264)))

# ---- SQ counters of the steady launches
with open("profiles/r06_mx_pmc_sq_summary.txt", "w") as out:
    out.write("SQ counters of the steady mx_kernel<7,4,2,4,512> launches at 1024 x 101 (8 388 608 rows; persistent blocks: grid 256 x 512), averages per launch;\n"
              "ACTIVE_* / WAIT_* / *_CYCLES of the SQ count quad-cycles (4 clocks) per wave or SIMD; SQ_VALU_MFMA_BUSY_CYCLES counts clocks per SIMD;\n"
              "GRBM_GUI_ACTIVE is summed over the 8 XCDs; two passes, tools/collect_profiles_r06.sh\n")
    vals, dur = {}, None
    for p in ("pmc_sq1", "pmc_sq2"):
        ls = steady(launches(p, KN))
        for k in sorted(ls[0]["c"]):
            vals[k] = sum(x["c"][k] for x in ls) / len(ls)
            out.write("%-32s %16.0f  (%d launches)\n" % (k, vals[k], len(ls)))
        if p == "pmc_sq2":
            dur = sum(x["dur_ns"] for x in ls) / len(ls) / 1e3
    if dur and "GRBM_GUI_ACTIVE" in vals:
        clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / dur
        out.write("average launch %.1f us under the profiler -> shader clock %.0f MHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)\n" % (dur, clk))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            out.write("matrix pipe busy: %.3f of the SIMD-cycles of the launch (1024 SIMDs x GRBM_GUI_ACTIVE / 8)\n"
                      % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0)))
print(open("profiles/r06_mx_pmc_sq_summary.txt").read())
