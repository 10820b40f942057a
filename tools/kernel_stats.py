"""Per-kernel launch statistics from a rocprofv3 --kernel-trace CSV, the frame's kernels only:
python tools/kernel_stats.py <dir with *_kernel_trace.csv>
Leaves out the one-workgroup launches (rm_set_scene renders one pixel to warm a freshly loaded kernel; rocprofv3's own
--stats table averages those in) and prints calls, average, minimum and maximum duration per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(path)))
by = defaultdict(list)
skipped = defaultdict(int)
for r in rows:
    size = int(r.get("Grid_Size") or (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
    wg = int(r.get("Workgroup_Size") or (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))
    if size <= wg and r["Kernel_Name"].startswith("rm_trace"):
        skipped[r["Kernel_Name"]] += 1
        continue
    by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':60s} {'calls':>6s} {'avg_ns':>12s} {'min_ns':>12s} {'max_ns':>12s} {'total_ms':>10s}")
for name, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:60]:60s} {len(d):6d} {sum(d) / len(d):12.0f} {min(d):12d} {max(d):12d} {sum(d) / 1e6:10.3f}")
for name, n in skipped.items():
    print(f"(left out: {n} one-workgroup warm-up launches of {name})")
