#!/bin/bash
# Same-box A/B of library builds (ab/lib_x.so ...; "-" is the in-tree build), kernel cache per leg: the default frame in
# flight and alone, and rank 0's share of an eight-way split alone and in flight (the frame's serial path).
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
    export RM_KERNEL_CACHE_DIR=/tmp/rm_aba_$(echo $L | tr '/' '_')
    timeout -k 10 120 python -u bench.py $B --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "lib=$L whole frame" || exit 1
    timeout -k 10 120 python -u bench.py $B --frames-in-flight 1 --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "lib=$L whole frame" || exit 1
    timeout -k 10 120 python -u bench.py $B --emulate-share 8 --frames-in-flight 1 --frames-per-gather 1 --steps 200 --warmup 20 ${BENCH_ARGS} 2>/dev/null | line "lib=$L eighth share" || exit 1
    timeout -k 10 120 python -u bench.py $B --emulate-share 8 --frames-in-flight 16 --frames-per-gather 1 --steps 400 --warmup 40 ${BENCH_ARGS} 2>/dev/null | line "lib=$L eighth share" || exit 1
done; done
