#!/bin/bash
# Sweep of compiler options for the kernels generated per scene ($RM_JIT_EXTRA, part of the code cache's key): ms per frame in
# flight and alone for each option set; a set the compiler refuses is reported and skipped.
#   WORKLOAD=default STEPS=200 bash tools/jit_flag_sweep.sh "<options>" "<options>" ...     ("" = the library's own options)
cd "${GRAFT_REPO_ROOT:-$PWD}"
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
W=${WORKLOAD:-default}; S=${STEPS:-200}
for rep in 1 2; do n=0; for leg in "$@"; do
    n=$((n+1))
    for f in 4 1; do
        RM_JIT_EXTRA="$leg" RM_KERNEL_CACHE_DIR=/tmp/rm_flags_$n timeout -k 10 200 python3 -u bench.py --workload $W $B --steps $S --warmup 5 --frames-in-flight $f 2>/tmp/rm_flags_err | python3 -c "
import json,sys
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); ok=True; print('[$leg] $W F $f: ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], flush=True)
if not ok: print('[$leg] $W F $f: no result:', open('/tmp/rm_flags_err').read()[-300:].replace(chr(10),' | '), flush=True)
"
    done
done; done
