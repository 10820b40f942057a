#!/bin/bash
# PMC passes of the default frame for a build-time knob setting: bash tools/pmc_ab.sh <tag> [VAR=val ...]
# (sets the variables, then runs the two SQ counter passes of tools/profile_round.sh on tools/one_frame.py)
set -e -o pipefail
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
WL=${WORKLOAD:-default}
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d "$OUT/pmc_sq1" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $WL 3 > "$OUT/pmc_sq1.log" 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d "$OUT/pmc_sq2" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $WL 3 > "$OUT/pmc_sq2.log" 2>&1
python3 "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
