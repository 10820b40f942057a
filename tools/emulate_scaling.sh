#!/bin/bash
# Single-GPU rehearsal of the N-GPU bench: rank 0's share of an N-way split at several frames-in-flight settings.
for N in 1 2 4 8; do for F in 1 4 8; do timeout -k 10 120 python -u bench.py --no-cpu-baseline --no-exhaustive-leg --emulate-share $N --frames-in-flight $F --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('N=$N F=$F: ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'Mrays/s x N', d['value'])
" || exit 1; done; done
