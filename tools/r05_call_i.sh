#!/bin/bash
# r05 GPU call I: op profiles of the secondary configs that have none of this round (configs[2] bf16_act, configs[3] NAFNet f32, latent score net fp16)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05i
mkdir -p "$OUT"
cd "$REPO"
timeout 400 python tools/op_profile.py 16 256 160 > "$OUT/op_profile_b16_256_bf16_act.txt" 2>&1
tail -1 "$OUT/op_profile_b16_256_bf16_act.txt"
timeout 400 python tools/op_profile.py 8 512 0 nafnet > "$OUT/op_profile_nafnet_b8_512.txt" 2>&1
tail -1 "$OUT/op_profile_nafnet_b8_512.txt"
timeout 400 python tools/op_profile.py 2 256 0 > "$OUT/op_profile_b2_256.txt" 2>&1
tail -1 "$OUT/op_profile_b2_256.txt"
