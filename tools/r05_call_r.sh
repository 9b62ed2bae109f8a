#!/bin/bash
# r05 GPU call R: the whole GPU suite and the driver's default bench line on the final tree (after the ragged-round tile rule)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05r
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -m gpu -q ) > "$OUT/pytest_gpu.txt" 2>&1
tail -8 "$OUT/pytest_gpu.txt" | cut -c1-250
( time timeout 1500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -3 "$OUT/bench_default.time"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05r/bench_default.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for s in d.get("secondary", []): print("  ", s.get("tag", "")[:60], s.get("value"), (s.get("roofline") or {}).get("frac"))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
