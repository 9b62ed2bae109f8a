#!/bin/bash
# r05 GPU call W: the whole GPU suite on the final tree (228 tests incl. the attention-sensitive goldens and block-level attention tests)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05w
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/pytest_gpu.txt" 2>&1
tail -8 "$OUT/pytest_gpu.txt" | cut -c1-250
