#!/bin/bash
# Round profile on the GPU box (run through gpurun): rocprofv3 kernel trace + the three PMC passes of bench.py's default
# workload, summarised into gpurun_out/<tag>/ (copy what should be judged into profiles/).
#   usage: bash tools/profile_round.sh <tag> [extra bench.py args]
set -u
TAG=${1:-prof}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary $*"
# 1. per-kernel durations (the same command bench.py times: 1 warm-up + 1 timed sampler call + the untimed profiled pass)
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $BENCH --steps 1 --warmup 1 > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name "*.db" | head -1)
[ -n "$DB" ] && python "$REPO/tools/rocprof_summary.py" "$DB" > "$OUT/bench_kernel_trace_stats.txt" 2>> "$OUT/kt.err"
# 2. HBM bytes: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots); 3 network evaluations are enough
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH --steps 1 --warmup 0 --T 3 --no-profile > /dev/null 2> "$OUT/pmc_$C.err"
done
F=$(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python "$REPO/tools/pmc_summary.py" "$F" "$W" 5 "$OUT/bench_pmc_hbm" "bench.py default workload + ($*)" > /dev/null 2> "$OUT/pmc_summary.err"
# 3. MFMA-pipe busy fraction per kernel
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o pmc -- $BENCH --steps 1 --warmup 0 --T 3 --no-profile > /dev/null 2> "$OUT/pmc_mfma.err"
M=$(find "$OUT/pmc_mfma" -name "*counter_collection.csv" | head -1)
[ -n "$M" ] && python "$REPO/tools/pmc_mfma_summary.py" "$M" "bench.py default workload ($*), 5 network evaluations (2-step setup call + T=3)" > "$OUT/bench_pmc_mfma.txt" 2>> "$OUT/pmc_mfma.err"
# keep the merged-back volume small: summaries only
rm -rf "$OUT/kt" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_mfma"
ls -la "$OUT"
