#!/bin/bash
# r05 GPU call C: group-sum probe, full GPU suite on the new build, headline / B=2 A/B against the r04 build, latent configs[4] with 1 / 2 / 4 concurrent sub-batches
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05c
mkdir -p "$OUT"
cd "$REPO"
timeout 60 tools/probe/group_sum_probe > "$OUT/group_sum_probe.txt" 2>&1; cat "$OUT/group_sum_probe.txt"
( time timeout 1200 python -m pytest tests -m gpu -q ) > "$OUT/pytest_gpu.txt" 2>&1
tail -25 "$OUT/pytest_gpu.txt" | cut -c1-200
for L in base new; do
  P=$REPO/image_restoration_sde_amd/libirsde_hip.so; [ $L = base ] && P=$REPO/image_restoration_sde_amd/libirsde_hip_base.so
  IRSDE_LIB_PATH=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > "$OUT/ab_$L.json" 2> "$OUT/ab_$L.err"
  IRSDE_LIB_PATH=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --batch 2 > "$OUT/ab_b2_$L.json" 2> "$OUT/ab_b2_$L.err"
done
IRSDE_LIB_PATH=$REPO/image_restoration_sde_amd/libirsde_hip_base.so timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-profile --steps 3 --warmup 1 --model latent --dtype fp16 --batch 64 > "$OUT/lat64_base.json" 2> "$OUT/lat64_base.err"
for N in 1 2 4; do
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-profile --steps 3 --warmup 1 --model latent --dtype fp16 --batch 64 > "$OUT/lat64_sub$N.json" 2> "$OUT/lat64_sub$N.err"
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-profile --steps 3 --warmup 1 --model latent --dtype fp16 --batch 32 > "$OUT/lat32_sub$N.json" 2> "$OUT/lat32_sub$N.err"
done
IRSDE_TUNING=1 IRSDE_SUBBATCHES=2 timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-profile --steps 3 --warmup 1 --model latent --dtype fp16 --batch 8 > "$OUT/lat8_sub2.json" 2> "$OUT/lat8_sub2.err"
grep -o '"value": *[0-9.]*' "$OUT"/*.json
