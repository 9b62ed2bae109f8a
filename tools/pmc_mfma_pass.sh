#!/bin/bash
# One rocprofv3 pass of SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over bench.py's workload (GPU box): MFMA-pipe busy fraction per kernel.
#   usage: bash tools/pmc_mfma_pass.sh <tag> [bench.py args]
set -u
TAG=${1:-mfma}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 0 --T 3 --no-profile "$@" > /dev/null 2> "$OUT/pmc_mfma.err"
M=$(find "$OUT/pmc_mfma" -name "*counter_collection.csv" | head -1)
[ -n "$M" ] && python "$REPO/tools/pmc_mfma_summary.py" "$M" "bench.py workload ($*), 5 network evaluations (2-step setup call + T=3)" > "$OUT/bench_pmc_mfma.txt"
rm -rf "$OUT/pmc_mfma"
cat "$OUT/bench_pmc_mfma.txt"
