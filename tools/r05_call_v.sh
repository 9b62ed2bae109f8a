#!/bin/bash
# r05 GPU call V: ConditionalUNet.forward vs the REAL reference with attention-sensitive weights (tests/golden/forward_attn.npz)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05v
mkdir -p "$OUT"
cd "$REPO"
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "attention_sensitive" ) > "$OUT/pytest_attn_sensitive.txt" 2>&1
grep -E "attention-sensitive|passed|failed|Error|assert" "$OUT/pytest_attn_sensitive.txt" | cut -c1-200
