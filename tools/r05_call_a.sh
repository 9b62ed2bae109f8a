#!/bin/bash
# r05 GPU call A: new parity tests + full GPU suite, default bench line (with secondaries), A/B of the headline against the r04 build,
# kernel traces of the per-GPU shard shapes (UNet B=2, latent B=8), and the 32x32-tile fused Winograd layout (r02 kernel, variant 80) under the TA/TD counters.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05a
mkdir -p "$OUT"
cd "$REPO"
( time timeout 900 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 600 "$OUT/bench_default.json"
# A/B: r04 build vs this build, headline only, same box
for L in base new; do
  P=$REPO/image_restoration_sde_amd/libirsde_hip.so; [ $L = base ] && P=$REPO/image_restoration_sde_amd/libirsde_hip_base.so
  IRSDE_LIB_PATH=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > "$OUT/ab_$L.json" 2> "$OUT/ab_$L.err"
  IRSDE_LIB_PATH=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --batch 2 > "$OUT/ab_b2_$L.json" 2> "$OUT/ab_b2_$L.err"
done
grep -o '"value": *[0-9.]*' "$OUT"/ab_*.json
# per-GPU shard shapes of the 8-GPU configs
bash tools/kernel_trace.sh r05a/kt_unet_b2 --batch 2 > /dev/null 2>&1
bash tools/kernel_trace.sh r05a/kt_latent_b8 --model latent --dtype fp16 --batch 8 > /dev/null 2>&1
bash tools/kernel_trace.sh r05a/kt_latent_b64 --model latent --dtype fp16 --batch 64 > /dev/null 2>&1
python tools/op_profile.py 2 256 0 > "$OUT/op_profile_b2.txt" 2>&1
# the 32-cout x 32-tile layout on v_mfma_f32_32x32x2_f32 (= r02's wino4_fused_kernel, bench_conv variant 80) vs production (430)
python tools/wino_one_layer.py 430,80 5 > "$OUT/wino_layout_time.txt" 2>&1
bash tools/pmc_ta_pass.sh r05a/ta "430,80" ta > /dev/null 2>&1
cp "$REPO/gpurun_out/r05a/ta/pmc_ta_ta.txt" "$OUT/" 2>/dev/null
rm -rf "$REPO"/gpurun_out/r05a/ta/q[0-9]
ls -la "$OUT"
