#!/bin/bash
# r05 GPU call L: the whole GPU suite on the tree with the 16-bit fused attention + halo2 kernels; NAFNet f32 tile-loop experiment for the plain 1x1 layers; configs[2] op profile
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05l
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/pytest_gpu.txt" 2>&1
tail -15 "$OUT/pytest_gpu.txt" | cut -c1-250
timeout 400 python tools/op_profile.py 16 256 160 > "$OUT/op_profile_b16_256_bf16_act.txt" 2>&1
tail -1 "$OUT/op_profile_b16_256_bf16_act.txt"
for Z in 1 2; do
  IRSDE_TUNING=1 IRSDE_ZLOOP_1X1=$Z timeout 400 python tools/op_profile.py 8 512 0 nafnet > "$OUT/op_profile_nafnet_zloop1x1_$Z.txt" 2>&1
  tail -1 "$OUT/op_profile_nafnet_zloop1x1_$Z.txt"
  grep "Cout=1024 Cin=512 k=1x1" "$OUT/op_profile_nafnet_zloop1x1_$Z.txt" | head -4 | cut -c1-170
done
