#!/bin/bash
# Vector memory path counters (address unit, L1, L2 requests) of a workload's frames: bash tools/pmc_mem.sh <tag> [VAR=val ...]
# ($WORKLOAD, default "synthetic"; three frames of tools/one_frame.py; sums over the rm_trace launches are printed)
set -e -o pipefail
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmcmem_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
WL=${WORKLOAD:-synthetic}
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d "$OUT/a" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $WL 3 > "$OUT/a.log" 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_READ_WAVEFRONTS_sum SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/b" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $WL 3 > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for part in ("a", "b"):
    sums = collections.defaultdict(float); launches = set()
    for f in glob.glob(sys.argv[1] + f"/{part}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == "rm_trace" and r.get("Grid_Size", "") != "64":
                sums[r["Counter_Name"]] += float(r["Counter_Value"]); launches.add(r["Dispatch_Id"])
    print(f"[{part}] rm_trace launches: {len(launches)}")
    for k in sorted(sums): print(f"  {k:38s} {sums[k]:.6g}")
PY
