#!/bin/bash
# r05 GPU call G: attention tests + op profile after the k/v staging change; shard-size table (256^2 and 512^2, B = 1..16 on one GPU); 512^2 kernel trace + op profile
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05g
mkdir -p "$OUT"
cd "$REPO"
( time timeout 900 python -m pytest tests -m gpu -q -k "attention or attn or sampler_256 or batch16 or forward_256 or split_mode or unet_forward" ) > "$OUT/pytest_sel.txt" 2>&1
tail -4 "$OUT/pytest_sel.txt" | cut -c1-200
timeout 600 python tools/op_profile.py 16 256 0 > "$OUT/op_profile_b16.txt" 2>&1
grep "linear_attention LayerNorm" "$OUT/op_profile_b16.txt"; tail -1 "$OUT/op_profile_b16.txt"
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1"
for N in 1 2 4 8 16; do timeout 600 $B --steps 2 --batch $N > "$OUT/shard256_b$N.json" 2> "$OUT/shard256_b$N.err"; done
for N in 1 2 4 8 16; do timeout 900 $B --steps 1 --size 512 --batch $N > "$OUT/shard512_b$N.json" 2> "$OUT/shard512_b$N.err"; done
grep -o '"value": *[0-9.]*' "$OUT"/shard*.json
bash tools/kernel_trace.sh r05g/kt_512 --size 512 --batch 16 > /dev/null 2>&1
timeout 600 python tools/op_profile.py 16 512 0 > "$OUT/op_profile_b16_512.txt" 2>&1
tail -1 "$OUT/op_profile_b16_512.txt"
