"""Turns two rocprofv3 counter runs of tools/one_frame.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE;
the two do not fit one pass) into a row of profiles/<round>/traffic.json, which bench.py reports as
roofline.traffic.

    python tools/collect_traffic.py --fetch DIR --write DIR --workload default --evaluator lattice \\
        --kernel-mode compiled --out profiles/r01/traffic.json

HBM bytes per trace launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB).  FETCH_SIZE counts the
128-byte requests of wide coalesced reads at 64 bytes on gfx950, hence the factor 2; WRITE_SIZE is exact for
the 8- and 16-byte-per-lane stores these kernels issue."""
import argparse
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument("--fetch", required=True)
ap.add_argument("--write", required=True)
ap.add_argument("--sq", default=None, help="directory of a pass with SQ_INSTS_VALU (optional): VALU wave-instructions per launch")
ap.add_argument("--workload", required=True)
ap.add_argument("--evaluator", required=True)
ap.add_argument("--kernel-mode", required=True)
ap.add_argument("--out", required=True)
a = ap.parse_args()


def per_launch(directory, counter):
    total, launches = 0.0, set()
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if row.get("Grid_Size") == "64":
                continue   # the one-pixel launch with which rm_set_scene warms a freshly loaded kernel
            if row["Counter_Name"] == counter and (name in ("rm_trace", "rm_trace_refill") or "trace_kernel" in name):
                total += float(row["Counter_Value"])
                launches.add(row["Dispatch_Id"])
    if not launches:
        raise SystemExit(f"no {counter} rows of the trace kernel under {directory}")
    return total / len(launches), len(launches)


fetch_kib, n_f = per_launch(a.fetch, "FETCH_SIZE")
write_kib, n_w = per_launch(a.write, "WRITE_SIZE")
row = {
    "workload": a.workload, "evaluator": a.evaluator, "kernel_mode": a.kernel_mode,
    "fetch_size_kib_per_launch": round(fetch_kib, 1), "write_size_kib_per_launch": round(write_kib, 1),
    "launches_averaged": [n_f, n_w],
    "hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024),
}
if a.sq:
    valu, n_v = per_launch(a.sq, "SQ_INSTS_VALU")
    row["valu_wave_insts_per_launch"] = int(valu)
    row["launches_averaged"].append(n_v)
rows = json.load(open(a.out)) if os.path.isfile(a.out) else []
rows = [r for r in rows if (r["workload"], r["evaluator"], r["kernel_mode"]) != (a.workload, a.evaluator, a.kernel_mode)]
rows.append(row)
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(rows, open(a.out, "w"), indent=1)
print(row)
