#!/bin/bash
# Same-box A/B of two builds of the library: bash tools/ab_lib.sh ab/lib_base.so [more libs]; "-" is the in-tree build.
cd "${GRAFT_REPO_ROOT:-$PWD}"
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'value', d['value'])
"; }
for rep in 1 2; do for L in "$@"; do
    if [ "$L" = "-" ]; then unset RM_LIB_PATH; else export RM_LIB_PATH=$PWD/$L; fi
    export RM_KERNEL_CACHE_DIR=/tmp/rm_abl_$(echo $L | tr '/' '_')
    timeout -k 10 120 python -u bench.py $B --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "lib=$L" || exit 1
done; done
