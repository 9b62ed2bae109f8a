#!/bin/bash
# r05 GPU call N: the per-GPU shard rates behind north_star's "256x256 and 512x512 batches at 1/2/4/8 GPUs" (strong scaling of a 16-image batch = 16 / 8 / 4 / 2 images per GPU;
# weak scaling = the 16-image line on every GPU), fp32 reverse_sde T=100, measured on ONE MI355X with the production path (graph replay)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r05n
mkdir -p "$OUT"
cd "$REPO"
B="python bench.py --no-cpu-baseline --no-secondary --no-profile --warmup 1"
for N in 16 8 4 2 1; do timeout 600 $B --steps 2 --batch $N > "$OUT/shard256_b$N.json" 2> "$OUT/shard256_b$N.err"; done
for N in 16 8 4 2 1; do timeout 900 $B --steps 1 --size 512 --batch $N > "$OUT/shard512_b$N.json" 2> "$OUT/shard512_b$N.err"; done
python - <<'PY'
import json, glob
rows = []
for size in (256, 512):
    for n in (16, 8, 4, 2, 1):
        try:
            d = json.load(open("gpurun_out/r05n/shard%d_b%d.json" % (size, n)))
            rows.append((size, n, d["value"], d["ms_per_step"]))
        except Exception as ex:
            rows.append((size, n, float("nan"), float("nan")))
with open("gpurun_out/r05n/shard_table.txt", "w") as f:
    f.write("# r05 call N (tools/r05_call_n.sh): fp32 reverse_sde T=100 on ONE MI355X, production path; images/s of a per-GPU shard of B images\n")
    f.write("# 'x GPUs' = what the sharded 16-image batch would deliver on that many GPUs IF every GPU ran this rate (dist.sample_shard has no collective inside the\n")
    f.write("# T loop; the final all_gather moves 12.6 MB at 256^2 / 50 MB at 512^2): DERIVED from the 1-GPU rate, not measured on a multi-GPU node\n")
    f.write("%6s %8s %12s %12s   %s\n" % ("size", "B/GPU", "images/s", "s per call", "strong scaling of a 16-image batch"))
    for size, n, v, ms in rows:
        g = 16 // n
        f.write("%6d %8d %12.3f %12.3f   %d GPU(s) x %d images -> %.2f images/s (derived)\n" % (size, n, v, ms / 1e3, g, n, g * v))
print(open("gpurun_out/r05n/shard_table.txt").read())
PY
