#!/bin/bash
# r05 GPU call Q: knob A/B in one box — 64-row tiles also for the 4.5-round T = 1024 component GEMMs of the headline (IRSDE_ZLOOP_RAGGED64=4096);
# the fused Winograd kernel from 512 tiles at the B = 2 shard (IRSDE_WINO_FUSED64_MINT=512)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05q
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1 --steps 2"
timeout 600 $B > "$OUT/head_default.json" 2> "$OUT/head_default.err"
IRSDE_TUNING=1 IRSDE_ZLOOP_RAGGED64=4096 timeout 600 $B > "$OUT/head_cut4096.json" 2> "$OUT/head_cut4096.err"
timeout 600 $B > "$OUT/head_default2.json" 2> "$OUT/head_default2.err"
timeout 600 $B --batch 2 > "$OUT/b2_default.json" 2> "$OUT/b2_default.err"
IRSDE_TUNING=1 IRSDE_WINO_FUSED64_MINT=512 timeout 600 $B --batch 2 > "$OUT/b2_mint512.json" 2> "$OUT/b2_mint512.err"
IRSDE_TUNING=1 IRSDE_WINO_FUSED64_MINT=512 timeout 600 $B --batch 4 > "$OUT/b4_mint512.json" 2> "$OUT/b4_mint512.err"
timeout 600 $B --batch 4 > "$OUT/b4_default.json" 2> "$OUT/b4_default.err"
grep -o '"value": *[0-9.]*' "$OUT"/*.json
IRSDE_TUNING=1 IRSDE_ZLOOP_RAGGED64=4096 timeout 400 python tools/op_profile.py 16 256 0 > "$OUT/op_profile_b16_cut4096.txt" 2>&1
grep "gemm x36" "$OUT/op_profile_b16_cut4096.txt" | cut -c1-150 | head -12; tail -1 "$OUT/op_profile_b16_cut4096.txt"
