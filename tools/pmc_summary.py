"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output) into HBM traffic per kernel group.
usage: python tools/pmc_summary.py <fetch_csv> <write_csv> <evals> <out_prefix>   -> <out_prefix>.txt / .json
Units and the gfx950 correction follow guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the raw counters are KiB;
FETCH_SIZE undercounts 16-byte-per-lane read streams by 2x on gfx950, so read bytes = 2 x FETCH_SIZE."""
import csv
import json
import sys
from collections import OrderedDict


def group(name):
    if "conv_igemm" in name or "gemm_zloop" in name or "conv3x3_halo" in name or "wino4_fused" in name or "conv3x3_narrow" in name or "gemm_split" in name:
        return "conv (conv_igemm / gemm_zloop / wino4_fused[64] / gemm_split / conv3x3_halo)"
    if "wino_" in name:
        return "wino_transform"
    if "attn_" in name:
        return "linear_attention"
    if "layernorm" in name:
        return "layernorm"
    if "irsde" in name:
        return "other irsde kernels"
    return "torch / runtime (bench harness)"


def load(path):
    agg = OrderedDict()
    for r in csv.DictReader(open(path)):
        g = group(r["Kernel_Name"])
        a = agg.setdefault(g, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
evals = int(sys.argv[3])
out = sys.argv[4]
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv), production plan:",
         "#   python bench.py --steps 1 --warmup 0 --T %d --no-cpu-baseline --no-profile   (%d network evaluations, B=16 256x256)" % (evals, evals),
         "# (the first evaluation of a fresh engine also uploads / packs nothing in the timed kernels; graph replay launches the same kernels)",
         "# raw counter unit = KiB; gfx950 correction (guides/MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE.", ""]
for title, agg in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    lines.append(title)
    lines.append("%-48s %10s %16s %18s" % ("kernel group", "launches", "raw GB (sum)", "raw MB / launch"))
    for g, (n, kib) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-48s %10d %16.3f %18.2f" % (g, n, kib * 1024 / 1e9, kib * 1024 / 1e6 / n))
    lines.append("")
ck, wk = "conv (conv_igemm / gemm_zloop / wino4_fused[64] / gemm_split / conv3x3_halo)", "wino_transform"
conv_launches = fetch[ck][0]
conv_bytes = (2 * fetch[ck][1] + write[ck][1]) * 1024
wino_bytes = (2 * fetch.get(wk, [0, 0.0])[1] + write.get(wk, [0, 0.0])[1]) * 1024
total = sum((2 * fetch[g][1] + write.get(g, [0, 0.0])[1]) * 1024 for g in fetch if "irsde" in g or g in (ck, wk, "layernorm", "linear_attention"))
lines += ["corrected HBM bytes (2 x FETCH + WRITE):",
          "  convolution kernels          %.2f GB per evaluation, %.1f MB per conv launch (%d launches per evaluation)" %
          (conv_bytes / evals / 1e9, conv_bytes / conv_launches / 1e6, conv_launches // evals),
          "  + Winograd transform kernels %.2f GB per evaluation" % (wino_bytes / evals / 1e9),
          "  whole network evaluation     %.2f GB" % (total / evals / 1e9)]
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_summary.py, see %s.txt" % out.split("/")[-1],
           "kernel": "convolution kernels (conv_igemm + gemm_zloop + wino4_fused) + wino transforms",
           "workload": "B=16 256x256 nf=64 depth=4, production plan",
           "traffic_bytes_per_launch": (conv_bytes + wino_bytes) / conv_launches,
           "conv_traffic_bytes_per_launch": conv_bytes / conv_launches,
           "bytes_per_evaluation": total / evals}, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
