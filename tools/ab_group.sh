#!/bin/bash
# Same-box A/B of the common box test in front of the root's sphere-bounded children (codegen.cpp: $RM_GROUP_TESTS:
# 0 none, 1 spheres and sphere-bounded lists, 2 box-bounded lists too), kernel cache per leg.
cd "${GRAFT_REPO_ROOT:-$PWD}"
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'value', d['value'])
"; }
for rep in 1 2; do for S in ${LEGS:-0 1 2}; do
    export RM_GROUP_TESTS=$S RM_KERNEL_CACHE_DIR=/tmp/rm_abg_$S
    timeout -k 10 120 python -u bench.py $B --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "group=$S" || exit 1
done; done
