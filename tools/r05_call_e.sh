#!/bin/bash
# r05 GPU call E: full GPU suite + default bench line + per-op profile (attention with the residual tile kept in registers)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05e
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1200 python -m pytest tests -m gpu -q ) > "$OUT/pytest_gpu.txt" 2>&1
tail -12 "$OUT/pytest_gpu.txt" | cut -c1-200
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 600 python tools/op_profile.py 16 256 0 > "$OUT/op_profile_b16.txt" 2>&1
grep "linear_attention LayerNorm" "$OUT/op_profile_b16.txt"; tail -1 "$OUT/op_profile_b16.txt"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05e/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for s in d["secondary"]: print(s["tag"], s.get("value"), s.get("ms_per_evaluation"))
PY
