#!/bin/bash
# r05 GPU call M: final evidence on the final tree — the driver's default bench line (secondaries, live traffic, cpu_baseline), the rocprofv3 round of the headline
# (kernel trace, HBM counters, MFMA counters), kernel traces of configs[2] (bf16_act ode) and configs[4] (latent fp16)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05m
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -3 "$OUT/bench_default.time"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05m/bench_default.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source","")[:100])
for s in d.get("secondary", []): print("  ", s.get("tag", s.get("workload", ""))[:60], s.get("value"), (s.get("roofline") or {}).get("frac"))
PY
bash tools/profile_round.sh r05_final2 > /dev/null 2>&1
bash tools/kernel_trace.sh r05_final2_bf16_act --dtype bf16_act --mode ode > /dev/null 2>&1
IRSDE_TUNING=1 IRSDE_SUBBATCHES=1 bash tools/kernel_trace.sh r05_final2_latent --model latent --dtype fp16 --batch 64 > /dev/null 2>&1
ls gpurun_out/r05_final2 gpurun_out/r05_final2_bf16_act gpurun_out/r05_final2_latent
head -14 gpurun_out/r05_final2/bench_kernel_trace_stats.txt | cut -c1-170
head -12 gpurun_out/r05_final2_bf16_act/bench_kernel_trace_stats.txt | cut -c1-170
