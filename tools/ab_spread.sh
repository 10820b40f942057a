#!/bin/bash
# Same-box A/B of the shadow rays spread over a tile part's idle lanes (trace_kernel.h: RM_SPREAD_SHADOWS), kernel cache
# off so that each leg measures its own tile order: whole default frame alone and in flight, and rank 0's share of
# an eight-way split alone (the frame's serial path).
cd "${GRAFT_REPO_ROOT:-$PWD}"
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'value', d['value'])
"; }
for rep in 1 2; do for S in 0 1; do
    export RM_JIT_EXTRA=-DRM_SPREAD_SHADOWS=$S RM_KERNEL_CACHE_DIR=/tmp/rm_ab_$S
    timeout -k 10 120 python -u bench.py $B --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "spread=$S whole frame" || exit 1
    timeout -k 10 120 python -u bench.py $B --frames-in-flight 1 --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "spread=$S whole frame" || exit 1
    timeout -k 10 120 python -u bench.py $B --emulate-share 8 --frames-in-flight 1 --frames-per-gather 1 --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "spread=$S eighth share" || exit 1
    timeout -k 10 120 python -u bench.py $B --emulate-share 8 --frames-in-flight 16 --frames-per-gather 1 --steps 400 --warmup 40 ${BENCH_ARGS} 2>/dev/null | line "spread=$S eighth share" || exit 1
done; done
