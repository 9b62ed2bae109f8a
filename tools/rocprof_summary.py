"""Summarise a rocprofv3 rocpd database (kernel trace) into text: per-kernel stats and, for the conv kernel,
per-launch-slot averages (the network evaluation is a fixed sequence of launches).
usage: python tools/rocprof_summary.py <results.db> [n_conv_per_eval] > profiles/xxx.txt"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (durations in us) from", sys.argv[1])
print("%-110s %8s %14s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-110s %8d %14.1f %10.1f %6.2f%%" % (name[:110], calls, total / 1e3 if total > 1e9 else total, avg / 1e3 if total > 1e9 else avg, pct))
rows = list(cur.execute("select name,start,duration,grid_x,grid_y,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size from kernels "
                        "where name like '%conv_igemm%' order by start"))
if rows:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 75
    print("\n# conv_igemm launches by slot within one network evaluation (%d conv launches per evaluation, %d evaluations)" % (n, len(rows) // n))
    agg = defaultdict(list)
    for i, r in enumerate(rows):
        agg[i % n].append(r)
    print("%4s %-14s %8s %6s %10s %6s %6s %8s" % ("slot", "tile", "grid_x", "grid_y", "avg_us", "vgpr", "sgpr", "lds"))
    for k in range(n):
        rs = agg[k]
        name = rs[0][0]
        tile = name[name.index("<") + 1:name.index(">")]
        avg = sum(r[2] for r in rs) / len(rs) / 1e3
        wg = rs[0][5]
        print("%4d %-14s %8d %6d %10.1f %6d %6d %8d" % (k, tile, rs[0][3] // wg, rs[0][4], avg, rs[0][6] + rs[0][7], rs[0][8], rs[0][9]))
