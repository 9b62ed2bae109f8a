#!/bin/bash
# r05 GPU call K: conv3x3_halo2_kernel (512 pixels x 128 channels per block): parity on forced small shapes, layer sweep old / new, bf16 network tests, op profile, bench rate
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05k
mkdir -p "$OUT"
cd "$REPO"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "halo2 or conv_kernel_bf16 or conv_kernel_fp16 or bf16_act_mode or fp16_mode" ) > "$OUT/pytest_sel.txt" 2>&1
grep -E "halo2|passed|failed|Error|error|assert" "$OUT/pytest_sel.txt" | cut -c1-250 | tail -20
SWEEP_VARIANTS=65,64,67,66 timeout 600 python tools/bf16_conv_sweep.py 16 3x3 > "$OUT/halo2_sweep.txt" 2>&1
SWEEP_VARIANTS=65,64,67,66 timeout 600 python tools/bf16_conv_sweep.py 16 up >> "$OUT/halo2_sweep.txt" 2>&1
cat "$OUT/halo2_sweep.txt" | cut -c1-200
( time timeout 900 python -m pytest tests/test_gpu_fullres.py -m gpu -q -x -s -k "fused_attention or reduced_precision" ) > "$OUT/pytest_fullres.txt" 2>&1
grep -E "fused attention|reverse_ode|passed|failed|Error|error" "$OUT/pytest_fullres.txt" | cut -c1-250 | tail -12
timeout 400 python tools/op_profile.py 16 256 160 > "$OUT/op_profile_b16_256_bf16_act.txt" 2>&1
tail -1 "$OUT/op_profile_b16_256_bf16_act.txt"
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1 --steps 2 --dtype bf16_act --mode ode > "$OUT/bench_bf16_act.json" 2> "$OUT/bench_bf16_act.err"
grep -o '"value": *[0-9.]*' "$OUT/bench_bf16_act.json"
