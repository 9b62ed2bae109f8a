#!/bin/bash
# r05 GPU call F: default bench with the live PMC traffic measurement; rocprofv3 round (kernel trace, HBM counters, MFMA counters) of the headline and of configs[4]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05f
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1200 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -3 "$OUT/bench_default.time"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/bench_default.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source","")[:120])
PY
bash tools/profile_round.sh r05_final > /dev/null 2>&1
IRSDE_TUNING=1 IRSDE_SUBBATCHES=1 bash tools/profile_round.sh r05_final_latent --model latent --dtype fp16 --batch 64 > /dev/null 2>&1
ls -la gpurun_out/r05_final gpurun_out/r05_final_latent
head -12 gpurun_out/r05_final/bench_kernel_trace_stats.txt | cut -c1-160
tail -12 gpurun_out/r05_final/bench_pmc_hbm.txt | head -8
