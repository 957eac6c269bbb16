"""Copy the rocprofv3 summaries tools/collect_profiles.sh left under gpurun_out/prof_r01/ into profiles/ (the tracked,
judged copies): kernel / domain statistics, the launch timeline of the last timed step, the two PMC passes and the
derived HBM-traffic JSON, and the bench line. Usage: publish_profiles.py <prefix, e.g. r01b_coarse> <steady grid> <rows>"""
import csv, glob, os, re, shutil, subprocess, sys

prefix, grid, rows = sys.argv[1], sys.argv[2], sys.argv[3]
src = "gpurun_out/prof_r01"
os.makedirs("profiles", exist_ok=True)


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)  # older runs may still lie around


shutil.copy(one("stats/*/*_kernel_stats.csv"), "profiles/%s_bench_kernel_stats.csv" % prefix)
shutil.copy(one("stats/*/*_domain_stats.csv"), "profiles/%s_bench_domain_stats.csv" % prefix)
shutil.copy(one("pmc_fetch/*/*_counter_collection.csv"), "profiles/%s_pmc_fetch_size_counter_collection.csv" % prefix)
shutil.copy(one("pmc_write/*/*_counter_collection.csv"), "profiles/%s_pmc_write_size_counter_collection.csv" % prefix)
line = [l for l in open(os.path.join(src, "bench_line.json")) if l.startswith("{")][-1]
open("profiles/%s_bench_line.json" % prefix, "w").write(line)


def short(n):
    n = re.sub(r"^void ", "", n)
    if "radix_sort" in n:
        return "rocprim radix_sort"
    m = re.match(r"(?:kgwas::)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+(?:<[^>]*>)?)", n)
    return m.group(1).replace(" ", "") if m else n[:40]


tr = sorted(csv.DictReader(open(one("stats/*/*_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
# the last step = everything from the last dense-start launch of the exact scorer on
starts = [i for i, r in enumerate(tr) if "score_mfma_kernel" in r["Kernel_Name"] and (i == 0 or "score_mfma_kernel" not in tr[i - 1]["Kernel_Name"])]
step = tr[starts[-1]:]
t0 = int(step[0]["Start_Timestamp"])
with open("profiles/%s_bench_launch_timeline.csv" % prefix, "w") as f:
    f.write("kernel,start_ms_since_dense_chunk,duration_ms,grid_size\n")
    for r in step:
        f.write("%s,%.3f,%.3f,%s\n" % (short(r["Kernel_Name"]), (int(r["Start_Timestamp"]) - t0) / 1e6,
                                       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Grid_Size_X"]))
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", "profiles/%s_pmc_fetch_size_counter_collection.csv" % prefix,
                       "profiles/%s_pmc_write_size_counter_collection.csv" % prefix, "coarse_kernel<7, 1>", grid, rows, "136",
                       "profiles/%s_pmc_hbm_traffic.json" % prefix])
