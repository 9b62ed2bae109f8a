"""Vector-memory path counters per kernel from rocprofv3 --pmc passes (CSV files given on the command line): mean per dispatch and, for cycle
counters, the fraction of the dispatch's GPU cycles.  *_sum counters add up the 256 CUs' units (divided by 256 here); GRBM_GUI_ACTIVE is per XCD-summed (/ 8).
usage: python tools/pmc_ta_summary.py <title> <counter_collection.csv> [...]"""
import csv, re, sys
from collections import OrderedDict, defaultdict
print("# " + sys.argv[1])
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
dur = defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"(wino4_fused(?:64[phst]?)?_kernel)<([^>]*)>", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + "<" + m.group(2).replace(" ", "") + "> grid " + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    print("\n%s   (%d dispatches, %.1f us, %.0f GPU cycles)" % (k, len(dur[k]), sum(dur[k]) / max(len(dur[k]), 1), cyc))
    for n in sorted(c):
        if n == "GRBM_GUI_ACTIVE":
            continue
        v = c[n]
        per_cu = v / 256.0 if n.endswith("_sum") or n.startswith("SQ_") else v
        frac = (" = %.3f of the kernel's cycles per CU" % (per_cu / cyc)) if cyc and ("CYCLES" in n or "BUSY" in n or "STALL" in n or "FULL" in n) else ""
        print("    %-40s %16.0f  per CU %12.0f%s" % (n, v, per_cu, frac))
