"""Copy the rocprofv3 summaries tools/collect_profiles_r05.sh left under gpurun_out/prof_r05/ into profiles/ (the tracked, judged
copies): kernel statistics (bench workload, one-column scans at 100 M and 1.2 G rows, kinship, the 2048 x 201 shape), PMC passes
(HBM traffic of mx_kernel, of mxs_kernel at 2048 x 201 and of the narrow filter at 1.2 G rows; SQ counters of both block-scaled
filter forms) and the bench lines. Usage: python tools/publish_profiles_r05.py [traffic]"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, ".")
src = "gpurun_out/prof_r05"
os.makedirs("profiles", exist_ok=True)


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


def counter_rows(path, kname):
    return [r for r in csv.DictReader(open(path)) if kname in r["Kernel_Name"]]


def steady(path, kname, counter, grid=None):
    """(mean value, launches, grid) of `counter` over the launches of the kernel with the given (default: the largest) grid."""
    v = [(float(r["Counter_Value"]), int(r["Grid_Size"])) for r in counter_rows(path, kname) if r["Counter_Name"] == counter]
    g = grid or max(x[1] for x in v)
    big = [x[0] for x in v if x[1] == g]
    return sum(big) / len(big), len(big), g


def total(path, kname, counter):
    v = [float(r["Counter_Value"]) for r in counter_rows(path, kname) if r["Counter_Name"] == counter]
    return sum(v), len(v)


def durations(path, kname, grid):
    d = []
    for r in csv.DictReader(open(path)):
        g = r.get("Grid_Size", r.get("Grid_Size_X"))
        if kname in r["Kernel_Name"] and (grid is None or int(g) == grid):
            d.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


NOTE = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/collect_profiles_r05.sh); counter values are KiB; on "
        "gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so HBM-side read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM "
        "section); WRITE_SIZE is taken as is.")
from bench import kernel_source_sha16

# ---- mx_kernel at 1024 x 101 (the line's roofline.traffic quotes this file when the live measurement is off)
KN, GRID, ROWS = "mx_kernel", 2048 * 512, 8388608
f = steady(one("pmc_fetch/*/*_counter_collection.csv"), KN, "FETCH_SIZE", GRID)
w = steady(one("pmc_write/*/*_counter_collection.csv"), KN, "WRITE_SIZE", GRID)
t = 2.0 * f[0] * 1024.0 + w[0] * 1024.0
json.dump({"kernel": KN, "kernel_source_sha16": kernel_source_sha16("score_mx.hip"),
           "fetch": {"counter": "FETCH_SIZE", "launches_averaged": f[1], "value_KiB_per_launch": f[0]},
           "write": {"counter": "WRITE_SIZE", "launches_averaged": w[1], "value_KiB_per_launch": w[0]},
           "rows_per_launch": ROWS, "algorithmic_bytes_per_launch": ROWS * 136, "traffic_bytes_per_launch": t,
           "traffic_bytes_per_row": t / ROWS, "traffic_over_algorithmic": t / (ROWS * 136),
           "note": NOTE + " kernel_source_sha16 = sha256(score_mx.hip + kernels.h + score_common.h)[:16] when the profile was taken: bench.py does not use it for another version of the kernel."},
          open("profiles/r05_mx_pmc_hbm_traffic.json", "w"), indent=1)
print("mx_kernel 1024 x 101: traffic / algorithmic = %.3f" % (t / (ROWS * 136)))
if "traffic" in sys.argv[1:]:
    sys.exit(0)

strip_stats(one("stats/*/*_kernel_stats.csv"), "profiles/r05_bench_kernel_stats.csv")
strip_stats(one("p1_stats/*/*_kernel_stats.csv"), "profiles/r05_p1scan_kernel_stats.csv")
strip_stats(one("p1l_stats/*/*_kernel_stats.csv"), "profiles/r05_p1scan_large_kernel_stats.csv")
strip_stats(one("kin_stats/*/*_kernel_stats.csv"), "profiles/r05_kinship_kernel_stats.csv")
strip_stats(one("c3_stats/*/*_kernel_stats.csv"), "profiles/r05_config4_kernel_stats.csv")
for name in ("bench_line", "config4_line", "config4_streaming_line", "shard250M_line", "shard250M_2threads_line"):
    line = [l for l in open(os.path.join(src, name + ".json")) if l.startswith("{")][-1]
    open("profiles/r05_%s.json" % name, "w").write(line)

# ---- the resident plan at 2048 x 201 (the default there): five LDS groups x 768 threads per row block of 4096 rows
fr = steady(one("pmc_c3r_fetch/*/*_counter_collection.csv"), "mx_kernel", "FETCH_SIZE")
wr = steady(one("pmc_c3r_write/*/*_counter_collection.csv"), "mx_kernel", "WRITE_SIZE", fr[2])
rowsr = fr[2] // 768 // 5 * 4096
tr = 2.0 * fr[0] * 1024.0 + wr[0] * 1024.0
RES = {"kernel": "mx_kernel", "grid_threads": fr[2], "rows_per_launch_upper_bound": rowsr, "traffic_bytes_per_launch": tr,
       "traffic_over_algorithmic_lower_bound": tr / (rowsr * 264)}
print("mx_kernel (resident, 5 LDS groups) 2048 x 201: traffic / algorithmic >= %.3f" % (tr / (rowsr * 264)))
# ---- the streaming filter at 2048 x 201 (KGWAS_MXS=2): the largest launches of the pass (row blocks of 4096 rows x 512 threads, one operand group)
KS = "mxs_kernel"
f3 = steady(one("pmc_c3_fetch/*/*_counter_collection.csv"), KS, "FETCH_SIZE")
w3 = steady(one("pmc_c3_write/*/*_counter_collection.csv"), KS, "WRITE_SIZE", f3[2])
rows3 = f3[2] // 512 * 4096
t3 = 2.0 * f3[0] * 1024.0 + w3[0] * 1024.0
json.dump({"kernel": KS, "kernel_source_sha16": kernel_source_sha16("score_mxs.hip"),
           "shape": "2048 samples x 201 columns: ONE operand group (two column groups of 7 tiles per block), every row loaded once",
           "fetch": {"counter": "FETCH_SIZE", "launches_averaged": f3[1], "value_KiB_per_launch": f3[0]},
           "write": {"counter": "WRITE_SIZE", "launches_averaged": w3[1], "value_KiB_per_launch": w3[0]},
           "grid_threads": f3[2], "rows_per_launch_upper_bound": rows3, "algorithmic_bytes_per_launch_upper_bound": rows3 * 264,
           "traffic_bytes_per_launch": t3, "traffic_over_algorithmic_lower_bound": t3 / (rows3 * 264),
           "resident_plan_same_run": RES,
           "note": NOTE + " Rows per launch from the grid (the last row block of a chunk may be short: an upper bound, the ratio a lower bound by less than 0.5 %)."},
          open("profiles/r05_mx_pmc_hbm_traffic_2048x201.json", "w"), indent=1)
print("mxs_kernel 2048 x 201: traffic / algorithmic >= %.3f" % (t3 / (rows3 * 264)))

# ---- the narrow filter over 1.2 G rows x 1024 samples, one column: ALL its launches of the passes / all the rows they read
KNW = "narrow"
passes = 2  # tools/p1_large_once.py 1200000000 1 runs a warm-up pass and one timed pass
fN = total(one("p1l_fetch/*/*_counter_collection.csv"), KNW + "_staged", "FETCH_SIZE")
wN = total(one("p1l_write/*/*_counter_collection.csv"), KNW + "_staged", "WRITE_SIZE")
rowsN = 1_200_000_000 * passes
tN = 2.0 * fN[0] * 1024.0 + wN[0] * 1024.0
json.dump({"kernel": "narrow_staged_kernel", "kernel_source_sha16": kernel_source_sha16("score_narrow.hip"),
           "workload": "1.2 G rows x 1024 samples (163 GB resident), one column, %d passes: every launch of the kernel" % passes,
           "fetch": {"counter": "FETCH_SIZE", "launches_summed": fN[1], "value_KiB_total": fN[0]},
           "write": {"counter": "WRITE_SIZE", "launches_summed": wN[1], "value_KiB_total": wN[0]},
           "rows_read": rowsN, "algorithmic_bytes": rowsN * 136, "traffic_bytes": tN, "traffic_bytes_per_row": tN / rowsN,
           "traffic_over_algorithmic": tN / (rowsN * 136), "note": NOTE + " The row loads are non-temporal."},
          open("profiles/r05_narrow_pmc_hbm_traffic.json", "w"), indent=1)
print("narrow_staged_kernel 1.2 G rows: traffic / algorithmic = %.3f" % (tN / (rowsN * 136)))


def sq_summary(out_path, title, p1, p2, kname, grid):
    with open(out_path, "w") as out:
        out.write(title)
        vals = {}
        for p in (p1, p2):
            path = one(p + "/*/*_counter_collection.csv")
            names = sorted(set(r["Counter_Name"] for r in counter_rows(path, kname)))
            for k in names:
                v, n, g = steady(path, kname, k, grid)
                out.write("%-32s %16.0f  (%d launches)\n" % (k, v, n))
                vals[k] = v
                grid = g
        d = durations(one(p2 + "/*/*_kernel_trace.csv"), kname, grid)
        if d and "GRBM_GUI_ACTIVE" in vals:
            us = sum(d) / len(d) / 1e3
            clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / us
            out.write("average launch %.1f us under the profiler -> shader clock %.0f MHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)\n" % (us, clk))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
                out.write("matrix pipe busy: %.3f of the SIMD-cycles of the launch (1024 SIMDs x GRBM_GUI_ACTIVE / 8)\n"
                          % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0)))
    print(open(out_path).read())


HEAD = ("ACTIVE_* / WAIT_* / *_CYCLES of the SQ count quad-cycles (4 clocks) per wave or SIMD; SQ_VALU_MFMA_BUSY_CYCLES counts clocks per SIMD;\n"
        "GRBM_GUI_ACTIVE is summed over the 8 XCDs; two passes, tools/collect_profiles_r05.sh\n")
sq_summary("profiles/r05_mx_pmc_sq_summary.txt", "SQ counters of the steady mx_kernel<7,4,2,4,512> launches at 1024 x 101 (8 388 608 rows, grid 2048 x 512), averages per launch;\n" + HEAD,
           "pmc_sq1", "pmc_sq2", "mx_kernel", 2048 * 512)
sq_summary("profiles/r05_mxs_pmc_sq_summary.txt", "SQ counters of the steady mxs_kernel<7,4,2,512,3> launches at 2048 x 201 (the largest grid of the pass: row blocks of 4096 rows x 512 threads), averages per launch;\n" + HEAD,
           "pmc_c3_sq1", "pmc_c3_sq2", "mxs_kernel", None)
