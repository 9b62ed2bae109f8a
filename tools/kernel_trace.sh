#!/bin/bash
# rocprofv3 kernel trace of bench.py's workload only (GPU box): per-kernel durations -> gpurun_out/<tag>/bench_kernel_trace_stats.txt
#   usage: bash tools/kernel_trace.sh <tag> [bench.py args]
set -u
TAG=${1:-kt}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
DB=$(find "$OUT/kt" -name "*.db" | head -1)
[ -n "$DB" ] && python "$REPO/tools/rocprof_summary.py" "$DB" > "$OUT/bench_kernel_trace_stats.txt" 2>> "$OUT/kt.err"
rm -rf "$OUT/kt"
head -30 "$OUT/bench_kernel_trace_stats.txt" | cut -c1-200
