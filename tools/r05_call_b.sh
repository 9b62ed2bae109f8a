#!/bin/bash
# r05 GPU call B: chain tap debug, attention kernels (DPP LayerNorm reduction, contiguous-tile staging) A/B against the r04 build, attention + sampler tests
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05b
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python tools/dbg_chain_taps.py 1 0 > "$OUT/dbg_chain_1_0.txt" 2>&1
timeout 300 python tools/dbg_chain_taps.py 3 1 > "$OUT/dbg_chain_3_1.txt" 2>&1
cat "$OUT/dbg_chain_1_0.txt"
( time timeout 900 python -m pytest tests -m gpu -q -k "attention or attn or sampler or forward_256 or batch16 or naf_chain or nafnet_levels or latent_hidden or golden or wino_fused64 or native_library" ) > "$OUT/pytest_sel.txt" 2>&1
tail -15 "$OUT/pytest_sel.txt"
for L in base new; do
  P=$REPO/image_restoration_sde_amd/libirsde_hip.so; [ $L = base ] && P=$REPO/image_restoration_sde_amd/libirsde_hip_base.so
  IRSDE_LIB_PATH=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > "$OUT/ab_$L.json" 2> "$OUT/ab_$L.err"
  IRSDE_LIB_PATH=$P timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --batch 2 > "$OUT/ab_b2_$L.json" 2> "$OUT/ab_b2_$L.err"
  IRSDE_LIB_PATH=$P timeout 600 python tools/op_profile.py 16 256 0 > "$OUT/op_profile_b16_$L.txt" 2>&1
done
grep -o '"value": *[0-9.]*' "$OUT"/ab_*.json
grep "linear_attention" "$OUT"/op_profile_b16_base.txt | head -12
grep "linear_attention" "$OUT"/op_profile_b16_new.txt | head -12
