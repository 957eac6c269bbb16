"""Copy the rocprofv3 summaries tools/collect_profiles_r02.sh left under gpurun_out/prof_r02/ into profiles/ (the
tracked, judged copies): kernel statistics of the bench workload, the one-column scan and the kinship accumulation, the
PMC passes (HBM traffic of coarse_kernel<7,1> and of the narrow filter, SQ counters of the steady coarse launch) and the
bench lines. Usage: python tools/publish_profiles_r02.py"""
import collections, csv, glob, json, os, shutil, subprocess, sys

src = "gpurun_out/prof_r02"
os.makedirs("profiles", exist_ok=True)


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)


def strip_stats(path, out, keep=25):
    """kernel_stats.csv with the rocprim template names shortened (they are kilobytes long)."""
    rows = list(csv.DictReader(open(path)))
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:keep]:
            n = r["Name"]
            if len(n) > 160:
                n = n[:60] + " ... " + n[-60:]
            w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])


strip_stats(one("stats/*/*_kernel_stats.csv"), "profiles/r02_bench_kernel_stats.csv")
strip_stats(one("p1_stats/*/*_kernel_stats.csv"), "profiles/r02_p1scan_kernel_stats.csv")
strip_stats(one("kin_stats/*/*_kernel_stats.csv"), "profiles/r02_kinship_kernel_stats.csv")
for name in ("bench_line", "config4_line"):
    line = [l for l in open(os.path.join(src, name + ".json")) if l.startswith("{")][-1]
    open("profiles/r02_%s.json" % name, "w").write(line)


def counters(path, kname, grid):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kname in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def traffic(fetch_csv, write_csv, kname, grid, rows, row_bytes, out, note):
    f = counters(fetch_csv, kname, grid)["FETCH_SIZE"]
    w = counters(write_csv, kname, grid)["WRITE_SIZE"]
    t = 2.0 * f[0] * 1024.0 + w[0] * 1024.0
    j = {"kernel": kname, "fetch": {"counter": "FETCH_SIZE", "launches_averaged": f[1], "value_KiB_per_launch": f[0]},
         "write": {"counter": "WRITE_SIZE", "launches_averaged": w[1], "value_KiB_per_launch": w[0]},
         "rows_per_launch": rows, "algorithmic_bytes_per_launch": rows * row_bytes, "traffic_bytes_per_launch": t,
         "traffic_bytes_per_row": t / rows, "traffic_over_algorithmic": t / (rows * row_bytes), "note": note}
    json.dump(j, open(out, "w"), indent=1)
    print(out, "traffic / algorithmic = %.3f" % j["traffic_over_algorithmic"])


note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/collect_profiles_r02.sh); counter "
        "values are KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so HBM-side read bytes = 2 * FETCH_SIZE * "
        "1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.")
# steady one-slice coarse launches: 8 388 608 rows, 2048 blocks of 512 threads
traffic(one("pmc_fetch/*/*_counter_collection.csv"), one("pmc_write/*/*_counter_collection.csv"), "coarse_kernel<7, 1>", 2048 * 512,
        8388608, 136, "profiles/r02_coarse_pmc_hbm_traffic.json", note)
# narrow filter, the 32 M-row launches of the one-column scan (768 rows per block of 256 threads)
g = None
for r in csv.DictReader(open(one("p1_fetch/*/*_counter_collection.csv"))):
    if "narrow_staged" in r["Kernel_Name"]:
        g = max(g or 0, int(r["Grid_Size"]))
rows_big = g // 256 * 768
traffic(one("p1_fetch/*/*_counter_collection.csv"), one("p1_write/*/*_counter_collection.csv"), "narrow_staged_kernel", g, rows_big, 136,
        "profiles/r02_narrow_pmc_hbm_traffic.json", note + " Largest launch of tools/one_column.py (about %d rows)." % rows_big)
with open("profiles/r02_coarse_pmc_sq_summary.txt", "w") as f:
    f.write("SQ counters of the steady coarse_kernel<7,1> launches (8 388 608 rows, grid 2048 x 512), averages per launch;\n"
            "*_CYCLES / ACTIVE_* / WAIT_* count quad-cycles (4 clocks); two passes, tools/collect_profiles_r02.sh\n")
    for p in ("pmc_sq1", "pmc_sq2"):
        for k, (v, n) in sorted(counters(one(p + "/*/*_counter_collection.csv"), "coarse_kernel<7, 1>", 2048 * 512).items()):
            f.write("%-32s %16.0f  (%d launches)\n" % (k, v, n))
print(open("profiles/r02_coarse_pmc_sq_summary.txt").read())

# kinship Gram kernel: SQ counters per launch (tools/pmc_kin.sh)
if os.path.exists(os.path.join(src, "kin_pmc_summary.txt")):
    with open("profiles/r02_kinship_pmc_sq_summary.txt", "w") as f:
        f.write("SQ counters of kin_gram_kernel (one 2^20-row chunk x 1135 accessions, 990 blocks of 128), averages per launch;\n"
                "*_CYCLES / ACTIVE_* / WAIT_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES cycles; GRBM_GUI_ACTIVE summed over 8 XCDs\n")
        for l in open(os.path.join(src, "kin_pmc_summary.txt")):
            if l.startswith(("SQ_", "GRBM", "TCC")):
                f.write(l)
    print(open("profiles/r02_kinship_pmc_sq_summary.txt").read())
