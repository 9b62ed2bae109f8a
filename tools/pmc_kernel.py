"""Per-dispatch means of every counter in a set of rocprofv3 --pmc CSVs, for the kernels whose name matches a regex.
usage: python tools/pmc_kernel.py <kernel-regex> <counter_collection.csv>...   (driver: tools/pmc_kernel.sh)"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict

rx = re.compile(sys.argv[1])
agg = OrderedDict()  # short kernel name -> counter -> [sum, n]
dur = defaultdict(lambda: [0.0, 0])
for path in sys.argv[2:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if not rx.search(name):
            continue
        short = re.sub(r"\(.*", "", name)[-90:]
        a = agg.setdefault(short, OrderedDict()).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            d = dur[short]
            d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            d[1] += 1
for short, counters in agg.items():
    print("%s   (mean duration under the counters: %.1f us)" % (short, dur[short][0] / max(dur[short][1], 1)))
    for c, (s, n) in counters.items():
        print("    %-34s %16.0f  per dispatch (%d dispatches)" % (c, s / n, n))
