#!/bin/bash
# N = 8 rehearsal on one GPU (rank 0's eighth share, every frame through a one-rank RCCL gather): frames in flight, frames
# per gather and hardware queues per process swept.  bash tools/gather_sweep.sh
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'G', c['frames_per_gather'], 'queues', d['gpu_max_hw_queues'], 'ms/frame', d['ms_per_step'], 'host issue', d['host_issue_ms_per_frame'], 'kernel alone', d['roofline']['kernel_ms'], 'Mrays/s x N', d['value'])
"; }
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
for Q in 4 8 16; do
export GPU_MAX_HW_QUEUES=$Q
timeout -k 10 120 python -u bench.py $B --emulate-share 1 --force-collective --steps 400 --warmup 40 2>/dev/null | line "N=1 gather" || exit 1
for F in 2 3 4 7 8 15 16; do
timeout -k 10 120 python -u bench.py $B --emulate-share 8 --force-collective --frames-in-flight $F --frames-per-gather 1 --steps 400 --warmup 40 2>/dev/null | line "N=8 gather" || exit 1
done
done
