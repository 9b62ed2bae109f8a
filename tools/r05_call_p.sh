#!/bin/bash
# r05 GPU call P: 64-row tiles for ragged rounds of the Winograd component GEMMs (IRSDE_ZLOOP_RAGGED64): shard rates B = 2 / 4 / 8 with the rule off / on (one box), op profile B = 2, plan-level tests
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05p
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1 --steps 2"
for N in 2 4 8; do
  IRSDE_TUNING=1 IRSDE_ZLOOP_RAGGED64=0 timeout 600 $B --batch $N > "$OUT/off_b$N.json" 2> "$OUT/off_b$N.err"
  timeout 600 $B --batch $N > "$OUT/on_b$N.json" 2> "$OUT/on_b$N.err"
done
grep -o '"value": *[0-9.]*' "$OUT"/off_b*.json "$OUT"/on_b*.json
timeout 400 python tools/op_profile.py 2 256 0 > "$OUT/op_profile_b2_256.txt" 2>&1
grep "gemm x36" "$OUT/op_profile_b2_256.txt" | cut -c1-150 | head -30; tail -1 "$OUT/op_profile_b2_256.txt"
( time timeout 900 python -m pytest tests -m gpu -q -k "batch or plan or wino or sampler_256 or forward" ) > "$OUT/pytest_sel.txt" 2>&1
tail -5 "$OUT/pytest_sel.txt" | cut -c1-200
