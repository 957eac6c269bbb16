"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into the HBM-traffic JSON that
bench.py quotes. Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring>
<grid size of the launches to average> <rows per such launch> <bytes per row> <out.json>"""
import csv, json, sys

fetch_csv, write_csv, kname, grid, rows, row_bytes, out = sys.argv[1:8]
grid, rows, row_bytes = int(grid), int(rows), int(row_bytes)


def avg(path, counter):
    vals, n_all = [], 0
    for r in csv.DictReader(open(path)):
        if kname in r["Kernel_Name"] and r["Counter_Name"] == counter:
            n_all += 1
            if int(r["Grid_Size"]) == grid:
                vals.append(float(r["Counter_Value"]))
    return n_all, vals


nf, f = avg(fetch_csv, "FETCH_SIZE")
nw, w = avg(write_csv, "WRITE_SIZE")
fk, wk = sum(f) / len(f), sum(w) / len(w)
traffic = 2.0 * fk * 1024.0 + wk * 1024.0
j = {
    "kernel": kname,
    "fetch": {"counter": "FETCH_SIZE", "launches": nf, "launches_averaged": len(f), "value_KiB_per_launch": fk},
    "write": {"counter": "WRITE_SIZE", "launches": nw, "launches_averaged": len(w), "value_KiB_per_launch": wk},
    "rows_per_launch": rows,
    "algorithmic_bytes_per_launch": rows * row_bytes,
    "traffic_bytes_per_launch": traffic,
    "traffic_bytes_per_row": traffic / rows,
    "traffic_over_algorithmic": traffic / (rows * row_bytes),
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --rows "
            "40000000 --steps 1 --warmup 0 --no-cpu-baseline`; launches of %d rows averaged. MI355X_MICROARCH.md (HBM): "
            "counter values are KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so HBM-side read bytes = "
            "2 * FETCH_SIZE * 1024 (calibrated there for 16 B/lane streaming reads; this kernel reads 8 B/lane, 64 "
            "contiguous bytes per row per instruction, so the factor is an upper bound); WRITE_SIZE is taken as is." % rows,
}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j, indent=1))
