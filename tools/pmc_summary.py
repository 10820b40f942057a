"""Text summary of the PMC passes tools/profile_round.sh collects (sums over the rm_trace launches of each pass):
python tools/pmc_summary.py gpurun_out/<dir> > profiles/<round>/pmc_default_compiled.txt"""
import csv
import os
import sys
from collections import defaultdict

base = sys.argv[1]
print("rocprofv3 --pmc passes over tools/one_frame.py default 3 (three frames: the first in natural tile order with tile timing,\n"
      "the others cost-ordered); sums over the three rm_trace launches unless noted.  Collected by tools/profile_round.sh.\n")
for name in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    path = os.path.join(base, name, "p_counter_collection.csv")
    if not os.path.isfile(path):
        continue
    sums, launches = defaultdict(float), set()
    rows = list(csv.DictReader(open(path)))
    # the frame's kernel: rm_trace, or rm_trace_refill when lane refill is on (rm_trace then only ran the one-pixel warm-up launch of rm_set_scene)
    kernel = "rm_trace_refill" if any(r["Kernel_Name"] == "rm_trace_refill" for r in rows) else "rm_trace"
    for row in rows:
        if row["Kernel_Name"] != kernel or (row.get("Grid_Size") or row.get("Grid_Size_X") or "") in ("64",):
            continue
        sums[row["Counter_Name"]] += float(row["Counter_Value"])
        launches.add(row["Dispatch_Id"])
    n = max(1, len(launches))
    print(f"[{name}] {kernel} launches: {len(launches)}")
    for k in sorted(sums):
        print(f"  {k:26s} {sums[k]:.6g}")
    if "SQ_THREAD_CYCLES_VALU" in sums and sums.get("SQ_ACTIVE_INST_VALU"):
        print(f"  -> VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {sums['SQ_THREAD_CYCLES_VALU'] / (64 * sums['SQ_ACTIVE_INST_VALU']):.3f}")
        print(f"  -> VALU wave-instructions per frame = {sums['SQ_INSTS_VALU'] / n:.4g}, SALU {sums['SQ_INSTS_SALU'] / n:.4g}, LDS {sums['SQ_INSTS_LDS'] / n:.4g}")
    if "SQ_WAIT_ANY" in sums:
        tot = sums["SQ_WAIT_ANY"] + sums["SQ_WAIT_INST_ANY"] + sums["SQ_ACTIVE_INST_ANY"]
        print(f"  -> of WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY: parked {sums['SQ_WAIT_ANY'] / tot:.3f}, issue-stalled {sums['SQ_WAIT_INST_ANY'] / tot:.3f}")
    print()
