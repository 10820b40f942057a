#!/bin/bash
# Sweep of code generation options for the per-scene kernels: bash tools/jit_sweep.sh [workload]   (on the GPU box)
# Each line: the options ($RM_JIT_EXTRA), Mrays/s, ms per frame with frames in flight, trace kernel alone.
cd "${GRAFT_REPO_ROOT:-$PWD}"
W=${1:-default}
export RM_KERNEL_CACHE_DIR=off
while IFS= read -r opts; do
    RM_JIT_EXTRA="$opts" timeout -k 10 200 python bench.py --workload $W --steps 200 --no-cpu-baseline --no-exhaustive-leg --no-every-ray-leg --no-host-leg --no-first-frame-leg 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('[$opts]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
except Exception as e: print('[$opts] failed')"
done <<'LIST'

-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-sched-strategy=max-memory-clause
-mllvm -align-loops=32
-mllvm -align-loops=64
-O2
-mllvm -amdgpu-early-inline-all=true
-mllvm -enable-post-misched=false
-mllvm -amdgpu-enable-max-ilp-scheduling-strategy
-mllvm -amdgpu-waitcnt-forcezero=0 -mllvm -amdgpu-disable-unclustered-high-rp-reschedule
LIST
