"""Per-chunk timeline of one headline step from a rocprofv3 kernel trace: for every chunk (a chunk_prep_kernel launch opens it)
the kernels' durations and the idle gaps between them.
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace -f csv -d gpurun_out/ktrace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-subrecords
    python tools/chunk_timeline.py gpurun_out/ktrace"""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void ", "").replace("kgwas::", "")
    short = short.split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
# the LAST step: from the last dense scorer launch on
idx = [i for i, r in enumerate(rows) if r[2] in ("score_mfma_kernel", "score_valu_kernel")]
start = idx[-1] if idx else 0
rows = rows[start:]
t0 = rows[0][0]
chunks, cur = [], None
for s, e, n in rows:
    if n == "chunk_prep_kernel":
        cur = []
        chunks.append(cur)
    if cur is not None:
        cur.append((s, e, n))
print("kernels before the first sparse chunk:")
for s, e, n in rows:
    if n == "chunk_prep_kernel": break
    print("  %8.1f us  %-28s %7.1f us" % ((s - t0) / 1e3, n, (e - s) / 1e3))
tot_busy = tot_gap = 0
prev_end = None
for ci, c in enumerate(chunks):
    busy = sum(e - s for s, e, n in c)
    span = c[-1][1] - c[0][0]
    gap_in = span - busy
    gap_before = (c[0][0] - prev_end) if prev_end else 0
    prev_end = max(e for s, e, n in c)
    tot_busy += busy; tot_gap += gap_in + max(gap_before, 0)
    per = collections.OrderedDict()
    for s, e, n in c:
        per[n] = per.get(n, 0) + (e - s)
    print("chunk %2d at %8.1f us: span %7.1f busy %7.1f gaps inside %6.1f before %6.1f | " % (ci, (c[0][0] - t0) / 1e3, span / 1e3, busy / 1e3, gap_in / 1e3, gap_before / 1e3) +
          " ".join("%s %.0f" % (k.replace("_kernel", ""), v / 1e3) for k, v in per.items()))
print("total busy %.2f ms, gaps %.2f ms, span %.2f ms" % (tot_busy / 1e6, tot_gap / 1e6, (rows[-1][1] - t0) / 1e6))
