#!/bin/bash
# Same-box A/B of environment knobs read when a scene's kernels are generated or built: bash tools/ab_env_alone.sh "VAR=val" "VAR=val2 RM_LIB_PATH=ab/x.so" ...
# (one quoted setting per leg, kernel cache per leg): the default frame in flight and alone, rank 0's eighth share alone and in flight.
cd "${GRAFT_REPO_ROOT:-$PWD}"
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'value', d['value'])
"; }
for rep in 1 2; do n=0; for leg in "$@"; do
    n=$((n+1))
    run() { env $leg RM_KERNEL_CACHE_DIR=/tmp/rm_abe_$n timeout -k 10 120 python -u bench.py $B "$@" ${BENCH_ARGS} 2>/dev/null; }
    run --steps 200 --warmup 20 | line "[$leg] whole frame" || exit 1
    run --frames-in-flight 1 --steps 200 --warmup 20 | line "[$leg] whole frame" || exit 1
    if [ -z "$NO_SHARE" ]; then
    run --emulate-share 8 --frames-in-flight 1 --frames-per-gather 1 --steps 200 --warmup 20 | line "[$leg] eighth share" || exit 1
    run --emulate-share 8 --frames-in-flight 16 --frames-per-gather 1 --steps 400 --warmup 40 | line "[$leg] eighth share" || exit 1
    fi
done; done
