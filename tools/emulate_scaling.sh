#!/bin/bash
# Single-GPU rehearsal of the N-GPU bench: rank 0's share of an N-way split, first without the exchange step at
# several frames-in-flight settings, then with bench.py's defaults and every frame sent through a one-rank RCCL
# gather (the collective's host and stream cost; what it cannot show is the transfer between GPUs).
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'G', c['frames_per_gather'], 'ms/frame', d['ms_per_step'], 'host issue', d['host_issue_ms_per_frame'], 'kernel alone', d['roofline']['kernel_ms'], 'Mrays/s x N', d['value'])
"; }
for N in 1 2 4 8; do for F in 1 4 8 16; do
    timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg --emulate-share $N --frames-in-flight $F --frames-per-gather 1 --steps 200 --warmup 20 2>/dev/null | line "N=$N no gather" || exit 1
done; done
for N in 1 2 4 8; do
    timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg --emulate-share $N --force-collective --steps 400 --warmup 40 2>/dev/null | line "N=$N one-rank RCCL gather" || exit 1
done
for N in 1 8; do
    timeout -k 10 200 python -u bench.py --workload uhd --no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg --emulate-share $N --force-collective --steps 20 --warmup 4 2>/dev/null | line "uhd N=$N one-rank RCCL gather" || exit 1
done
