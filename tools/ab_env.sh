#!/bin/bash
# A/B of build-time knobs read from the environment: bash tools/ab_env.sh "VAR=val VAR2=val" "..." (one quoted setting per leg)
cd "${GRAFT_REPO_ROOT:-$PWD}"
for leg in "$@"; do
    for rep in 1 2; do
        env $leg timeout -k 10 200 python bench.py --steps 200 --no-cpu-baseline --no-exhaustive-leg --no-every-ray-leg ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$leg', 'value', d['value'], 'ms', d['ms_per_step'], 'alone', d['roofline']['kernel_ms'])" || exit 1
    done
done
