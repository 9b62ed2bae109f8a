#!/bin/bash
# r05 GPU call D: independent per-part step graphs (sub-batches on separate streams): latent configs[4] sweep, UNet small-batch sweep; chain / attention tests
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05d
mkdir -p "$OUT"
cd "$REPO"
timeout 60 tools/probe/group_sum_probe > "$OUT/group_sum_probe.txt" 2>&1; tail -2 "$OUT/group_sum_probe.txt"
( time timeout 900 python -m pytest tests -m gpu -q -k "naf or latent or subbatch or attention or attn or sampler_256 or batch16 or bokeh" ) > "$OUT/pytest_sel.txt" 2>&1
tail -8 "$OUT/pytest_sel.txt" | cut -c1-200
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --steps 3 --warmup 1"
for N in 1 2 4; do
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 $B --model latent --dtype fp16 --batch 64 > "$OUT/lat64_sub$N.json" 2> "$OUT/lat64_sub$N.err"
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 $B --model latent --dtype fp16 --batch 32 > "$OUT/lat32_sub$N.json" 2> "$OUT/lat32_sub$N.err"
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 $B --model latent --dtype fp16 --batch 16 > "$OUT/lat16_sub$N.json" 2> "$OUT/lat16_sub$N.err"
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 $B --model latent --dtype fp16 --batch 8 > "$OUT/lat08_sub$N.json" 2> "$OUT/lat08_sub$N.err"
done
for N in 1 2; do
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 $B --batch 2 > "$OUT/unet02_sub$N.json" 2> "$OUT/unet02_sub$N.err"
  IRSDE_TUNING=1 IRSDE_SUBBATCHES=$N timeout 600 $B --batch 4 > "$OUT/unet04_sub$N.json" 2> "$OUT/unet04_sub$N.err"
done
IRSDE_TUNING=1 IRSDE_SUBBATCHES=4 timeout 600 $B --batch 4 > "$OUT/unet04_sub4.json" 2> "$OUT/unet04_sub4.err"
timeout 600 $B --steps 2 > "$OUT/unet16.json" 2> "$OUT/unet16.err"
IRSDE_TUNING=1 IRSDE_SUBBATCHES=2 timeout 600 $B --steps 2 > "$OUT/unet16_sub2.json" 2> "$OUT/unet16_sub2.err"
IRSDE_TUNING=1 IRSDE_SUBBATCHES=2 timeout 600 $B --steps 1 --model nafnet --batch 8 --size 512 --T 100 > "$OUT/naf8x512_sub2.json" 2> "$OUT/naf8x512_sub2.err"
IRSDE_TUNING=1 IRSDE_SUBBATCHES=1 timeout 600 $B --steps 1 --model nafnet --batch 8 --size 512 --T 100 > "$OUT/naf8x512_sub1.json" 2> "$OUT/naf8x512_sub1.err"
grep -o '"value": *[0-9.]*' "$OUT"/*.json
