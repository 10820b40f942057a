for rep in 1 2; do for v in 0 1; do
RM_DIRECT_HOST=$v timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-exhaustive-leg --no-first-frame-leg --no-every-ray-leg ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); h=d['host_image']; print('RM_DIRECT_HOST=$v', 'alone host', h['frame_alone_ms_host'], 'alone dev', h['frame_alone_ms'], 'pageable', h['frame_alone_ms_host_pageable'], 'in flight host', h['frame_ms_host'], 'equal', h['host_frames_equal_device_frame'])
" || exit 1
done; done
