#!/bin/bash
# r05 GPU call S: plan-choice A/B in one box (bench.py --engine-flags, measurement only): the per-image NAFBlock chain vs the per-layer path at the 8-GPU shard of configs[4]
# (8 images) and at 64 images ("or fall back per layer if that measures faster"); configs[2] with / without the fused 16-bit attention kernels
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05s
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1 --steps 3"
for N in 8 64; do
  timeout 600 $B --model latent --dtype fp16 --batch $N > "$OUT/latent_b${N}_chain.json" 2> "$OUT/latent_b${N}_chain.err"
  timeout 600 $B --model latent --dtype fp16 --batch $N --engine-flags 8192 > "$OUT/latent_b${N}_layers.json" 2> "$OUT/latent_b${N}_layers.err"
done
timeout 600 $B --dtype bf16_act --mode ode > "$OUT/bf16act_fused.json" 2> "$OUT/bf16act_fused.err"
timeout 600 $B --dtype bf16_act --mode ode --engine-flags 4096 > "$OUT/bf16act_tensorplan.json" 2> "$OUT/bf16act_tensorplan.err"
grep -o '"value": *[0-9.]*' "$OUT"/*.json
