#!/bin/bash
# Collects the round's profiles on the GPU box: bash tools/profile_round.sh <out dir under gpurun_out>
# (bench lines of every workload, rocprofv3 kernel statistics of the default bench, TCC traffic counters).
# $PART=1: bench lines, rehearsal and lane utilisation only; $PART=2: rocprofv3 passes only (each half fits one GPU call).
set -e -o pipefail
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/${1:-profile}
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$OUT"
cd "$REPO"
if [ "${PART:-1}" != 2 ]; then
timeout -k 10 300 python3 -u bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
bash tools/emulate_scaling.sh > "$OUT/emulate_scaling.txt" 2>&1
timeout -k 10 300 python3 tools/march_util.py default uhd menger2 closeup6 synthetic synthetic_wide > "$OUT/march_util.txt" 2>/dev/null
for w in menger2 uhd closeup6 synthetic synthetic_wide; do
    steps=10; [ $w = menger2 ] && steps=200   # a menger2 frame is 0.1 ms: few frames time the launch, not the frame
    timeout -k 10 300 python3 -u bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
fi
[ "${PART:-2}" = 1 ] && { echo done; exit 0; }
cd /tmp
export TMPDIR=/tmp
# one frame in flight here, so that every rm_trace launch runs alone and the statistics' average duration is
# the kernel's own (with several frames in flight the spans of co-scheduled kernels stretch)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o default --output-format csv -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --frames-in-flight 1 --no-cpu-baseline --no-exhaustive-leg --no-every-ray-leg --no-host-leg --no-first-frame-leg > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats_f4" -o default --output-format csv -- python3 "$REPO/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-every-ray-leg --no-host-leg --no-first-frame-leg > "$OUT/stats_f4_bench.json" 2> "$OUT/stats_f4.err"
python3 "$REPO/tools/kernel_stats.py" "$OUT/stats" > "$OUT/kernel_stats_f1.txt"
python3 "$REPO/tools/kernel_stats.py" "$OUT/stats_f4" > "$OUT/kernel_stats_f4.txt"
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" default 3 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" default 3 > "$OUT/pmc_write.log" 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d "$OUT/pmc_sq1" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" default 3 > "$OUT/pmc_sq1.log" 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d "$OUT/pmc_sq2" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" default 3 > "$OUT/pmc_sq2.log" 2>&1
python3 "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_default_compiled.txt"
python3 "$REPO/tools/collect_traffic.py" --fetch "$OUT/pmc_fetch" --write "$OUT/pmc_write" --sq "$OUT/pmc_sq1" --workload default --evaluator lattice --kernel-mode compiled --out "$OUT/traffic.json"
for w in menger2 uhd closeup6 synthetic; do
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch_$w" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $w 3 > "$OUT/pmc_fetch_$w.log" 2>&1
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write_$w" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $w 3 > "$OUT/pmc_write_$w.log" 2>&1
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d "$OUT/pmc_sq1_$w" -o p --output-format csv -- python3 "$REPO/tools/one_frame.py" $w 3 > "$OUT/pmc_sq1_$w.log" 2>&1
    python3 "$REPO/tools/collect_traffic.py" --fetch "$OUT/pmc_fetch_$w" --write "$OUT/pmc_write_$w" --sq "$OUT/pmc_sq1_$w" --workload $w --evaluator lattice --kernel-mode compiled --out "$OUT/traffic.json"
done
echo done
