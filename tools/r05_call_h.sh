#!/bin/bash
# r05 GPU call H: 16 x 512^2 with the fused Winograd layers on batch slices (tensors beyond 2 GiB): test, shard rates, op profile, kernel trace
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05h
mkdir -p "$OUT"
cd "$REPO"
( time timeout 900 python -m pytest tests -m gpu -q -k "512 or split_mode or batch16" ) > "$OUT/pytest_sel.txt" 2>&1
tail -6 "$OUT/pytest_sel.txt" | cut -c1-220
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1"
for N in 8 16; do timeout 900 $B --steps 1 --size 512 --batch $N > "$OUT/shard512_b$N.json" 2> "$OUT/shard512_b$N.err"; done
timeout 900 $B --steps 1 --size 512 --batch 16 --dtype fp32_split_f16 > "$OUT/shard512_b16_split.json" 2> "$OUT/shard512_b16_split.err"
grep -o '"value": *[0-9.]*' "$OUT"/shard*.json
bash tools/kernel_trace.sh r05h/kt_512 --size 512 --batch 16 > /dev/null 2>&1
timeout 600 python tools/op_profile.py 16 512 0 > "$OUT/op_profile_b16_512.txt" 2>&1
tail -1 "$OUT/op_profile_b16_512.txt"
