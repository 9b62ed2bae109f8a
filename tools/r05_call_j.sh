#!/bin/bash
# r05 GPU call J: fused attention kernels with 16-bit projection operands (bf16 / bf16_act / fp16): new test, the reduced-precision trajectory test, op profile, bench rate
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05j
mkdir -p "$OUT"
cd "$REPO"
( time timeout 900 python -m pytest tests/test_gpu_fullres.py -m gpu -q -x -s -k "fused_attention or reduced_precision" ) > "$OUT/pytest_sel.txt" 2>&1
grep -E "fused attention|reverse_ode|passed|failed|Error|error" "$OUT/pytest_sel.txt" | cut -c1-250 | tail -20
timeout 400 python tools/op_profile.py 16 256 160 > "$OUT/op_profile_b16_256_bf16_act.txt" 2>&1
grep "linear_attention" "$OUT/op_profile_b16_256_bf16_act.txt" | cut -c1-160; tail -1 "$OUT/op_profile_b16_256_bf16_act.txt"
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1 --steps 2 --dtype bf16_act --mode ode > "$OUT/bench_bf16_act.json" 2> "$OUT/bench_bf16_act.err"
grep -o '"value": *[0-9.]*' "$OUT/bench_bf16_act.json"
