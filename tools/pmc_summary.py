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


FAMILIES = ("wino4_fused64", "wino4_fused", "gemm_zloop", "gemm_split2i", "gemm_split", "conv_igemm", "conv3x3_halo", "conv3x3_narrow", "wino_input",
            "wino_output", "attn_kv_ctx", "attn_q_out_fused", "attn_", "layernorm")


def family(name):
    return next((f for f in FAMILIES if f in name), None)


OUTSIDE = {}   # counter -> [launches, KiB] of the irsde kernels OUTSIDE the sampler steps (latent encode / decode, layout kernels): reported apart


def load(path):
    """-> ({group: [launches, KiB]}, [(family, grid size, KiB) of every irsde kernel launch, in dispatch order]).
    r05: only launches INSIDE a sampler step count (from a step_begin_kernel through the sde_update_kernel that ends the step) — the latent pipeline's
    encode / decode run once per call and used to be amortised over the evaluations; they are summed into OUTSIDE instead.
    The number of evaluations is the number of step_begin launches found (EVALS_FOUND)."""
    global EVALS_FOUND
    agg, seq = OrderedDict(), []
    rows = list(csv.DictReader(open(path)))
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    has_steps = any("step_begin_kernel" in r["Kernel_Name"] for r in rows)
    in_step, nsteps, out = False, 0, [0, 0.0]
    for r in rows:
        name = r["Kernel_Name"]
        if has_steps:
            if "step_begin_kernel" in name:
                in_step, nsteps = True, nsteps + 1
            if not in_step:
                if "irsde" in name:
                    out[0] += 1
                    out[1] += float(r["Counter_Value"])
                continue
            if "sde_update_kernel" in name:
                in_step = False   # (this row still belongs to the step)
        g = group(r["Kernel_Name"])
        a = agg.setdefault(g, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        f = family(r["Kernel_Name"])
        if f:
            seq.append((f, r.get("Grid_Size", "?"), float(r["Counter_Value"])))
    OUTSIDE[path] = out
    if has_steps:
        EVALS_FOUND = nsteps
    return agg, seq


EVALS_FOUND = 0


(fetch, fseq), (write, wseq) = load(sys.argv[1]), load(sys.argv[2])
evals = EVALS_FOUND or int(sys.argv[3])
out = sys.argv[4]
workload = sys.argv[5] if len(sys.argv) > 5 else "B=16 256x256 nf=64 depth=4, production plan"
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv), production plan:",
         "#   bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-profile [workload arguments]   (%d network evaluations counted; workload: %s)" % (evals, workload),
         "# only launches inside sampler steps (step_begin .. sde_update) are counted per evaluation; kernels outside them: see the last lines",
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
# per kernel family, and the launches of the LAST evaluation one by one (dispatch order = plan order: compare with irsde_op_profile's lines)
if len(fseq) == len(wseq) and all(a[0] == b[0] for a, b in zip(fseq, wseq)):
    fam = OrderedDict()
    for (f, _, fk), (_, _, wk_) in zip(fseq, wseq):
        a = fam.setdefault(f, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += 2 * fk * 1024
        a[2] += wk_ * 1024
    lines += ["", "per kernel family (corrected bytes per evaluation):", "%-22s %9s %12s %12s" % ("family", "launches", "read GB", "write GB")]
    for f, (n, rb, wb) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        lines.append("%-22s %9d %12.3f %12.3f" % (f, n // evals, rb / evals / 1e9, wb / evals / 1e9))
    per = len(fseq) // evals
    lines += ["", "launches of the last evaluation (dispatch order):", "%4s %-22s %12s %12s %12s" % ("#", "family", "grid", "read MB", "write MB")]
    for i, ((f, grid, fk), (_, _, wk_)) in enumerate(zip(fseq[-per:], wseq[-per:])):
        lines.append("%4d %-22s %12s %12.1f %12.1f" % (i, f, grid, 2 * fk * 1024 / 1e6, wk_ * 1024 / 1e6))
else:
    lines += ["", "(per-launch table skipped: the two passes did not record the same launch sequence)"]
fo, wo = OUTSIDE.get(sys.argv[1], [0, 0.0]), OUTSIDE.get(sys.argv[2], [0, 0.0])
if fo[0]:
    lines += ["", "outside the sampler steps (once per call: latent encode / decode, layout and harness kernels): %d launches, %.2f GB corrected (2 x FETCH + WRITE)"
              % (fo[0], (2 * fo[1] + wo[1]) * 1024 / 1e9)]
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_summary.py, see %s.txt" % out.split("/")[-1],
           "kernel": "convolution kernels (conv_igemm + gemm_zloop + wino4_fused) + wino transforms",
           "workload": workload,
           "traffic_bytes_per_launch": (conv_bytes + wino_bytes) / conv_launches,
           "conv_traffic_bytes_per_launch": conv_bytes / conv_launches,
           "bytes_per_evaluation": total / evals}, open(out + ".json", "w"), indent=1)
print("\n".join(lines[:40]))
