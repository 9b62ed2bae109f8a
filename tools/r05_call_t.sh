#!/bin/bash
# r05 GPU call T: block-level attention parity (fp32 and the 16-bit operand modes) against the oracle with attention-sensitive weights
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05t
mkdir -p "$OUT"
cd "$REPO"
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "fused_attention_block" ) > "$OUT/pytest_attn_block.txt" 2>&1
grep -E "C=|passed|failed|Error|assert" "$OUT/pytest_attn_block.txt" | cut -c1-220
