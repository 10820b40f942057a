#!/bin/bash
# Same-box A/B of where a scene's kernels are compiled: by the program rm_jit beside the library (a process of its own, the
# system's hiprtc) or inside the calling process ($RM_JIT_HELPER=off; under bench.py that is the hiprtc PyTorch bundles).
# A fresh cache directory per leg.  bash tools/ab_jit_helper.sh [workloads]
set -e -o pipefail
cd "${GRAFT_REPO_ROOT:-$PWD}"
F="--no-cpu-baseline --no-host-leg --no-every-ray-leg --no-exhaustive-leg"
for rep in 1 2; do
for w in ${@:-default uhd closeup6 synthetic synthetic_wide menger2}; do
  steps=12; [ $w = default ] && steps=200; [ $w = menger2 ] && steps=200
  for leg in in_process rm_jit; do
    export RM_KERNEL_CACHE_DIR=/tmp/rmj_${leg}_${w}_$rep
    if [ $leg = in_process ]; then export RM_JIT_HELPER=off; else unset RM_JIT_HELPER; fi
    timeout -k 10 300 python3 -u bench.py --workload $w --steps $steps --warmup 3 $F 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); ff=d.get('first_frame') or {}
        print('$w $leg: ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'frame alone', d.get('frame_alone_device_ms'), 'compile_ms', d.get('compile_ms'), 'first frame', {k: v for k, v in ff.items() if k != 'note'}, flush=True)
"
  done
done
done
