"""MFMA-pipe utilisation per kernel group from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (CSV).
usage: python tools/pmc_mfma_summary.py <counter_collection.csv> <title> >> profiles/xxx.txt
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)  (as in profiles/r01_conv_pmc_mfma.txt)."""
import csv
import re
import sys
from collections import OrderedDict

rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Dispatch_Id"], r["Kernel_Name"])
    d = rows.setdefault(key, {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])


def group(name):
    m = re.search(r"(gemm_zloop_kernel|conv_igemm_kernel|conv3x3_halo2_kernel|conv3x3_halo_bf16_kernel|wino4_fused64p_kernel|wino4_fused64t_kernel|wino4_fused64h_kernel|wino4_fused64_kernel|wino4_fused_kernel|naf_chain_kernel|dwconv_gate_kernel|gemm_split2i_kernel|gemm_split_kernel|wino_input_split_kernel|wino_input_kernel|wino_output_kernel|"
                  r"layernorm_kernel|attn_\w+_kernel|conv3x3_narrow_kernel)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


agg = OrderedDict()
for (_, name), d in rows.items():
    g = group(name)
    if g is None or "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    a = agg.setdefault(g, [0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += d["dur"]
    a[2] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
    a[3] += d["GRBM_GUI_ACTIVE"]
print("# " + sys.argv[2])
print("%-78s %8s %12s %10s" % ("kernel", "launches", "total_us", "mfma_util"))
for g, (n, dur, busy, act) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    util = busy / (1024.0 * act / 8.0) if act else 0.0
    print("%-78s %8d %12.1f %10.3f" % (g[:78], n, dur, util))
tb = sum(v[2] for v in agg.values())
ta = sum(v[3] for v in agg.values())
print("%-78s %8s %12.1f %10.3f" % ("all of the above", "", sum(v[1] for v in agg.values()), tb / (1024.0 * ta / 8.0)))
print()
