cd "${GRAFT_REPO_ROOT:-$PWD}"
B="--no-cpu-baseline --no-exhaustive-leg --no-host-leg --no-first-frame-leg --no-every-ray-leg"
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('$1: F', c['frames_in_flight'], 'ms/frame', d['ms_per_step'], 'kernel alone', d['roofline']['kernel_ms'], 'value', d['value'])
"; }
for S in 16,32 32,64 64,64 64,128 128,256 16,128; do
    export RM_SPLIT_TILES=$S RM_KERNEL_CACHE_DIR=/tmp/rm_split_$S
    timeout -k 10 120 python -u bench.py $B --steps 200 --warmup 20 2>/dev/null | line "split=$S whole frame" || exit 1
    timeout -k 10 120 python -u bench.py $B --frames-in-flight 1 --steps 200 --warmup 20 2>/dev/null | line "split=$S whole frame" || exit 1
    timeout -k 10 120 python -u bench.py $B --emulate-share 8 --frames-in-flight 1 --frames-per-gather 1 --steps 200 --warmup 20 2>/dev/null | line "split=$S eighth share" || exit 1
done
