"""Summarise a rocprofv3 rocpd database (kernel trace) into text: per-kernel stats and per (kernel, grid) launch classes.
usage: python tools/rocprof_summary.py <results.db> > profiles/xxx.txt"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (durations in us) from", sys.argv[1])
print("%-110s %8s %14s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-110s %8d %14.1f %10.1f %6.2f%%" % (name[:110], calls, total / 1e3 if total > 1e9 else total, avg / 1e3 if total > 1e9 else avg, pct))
# per (kernel, grid) launch classes: robust against plan changes (r02's "slot" table assumed 75 conv_igemm launches per
# evaluation and was meaningless once the plan changed)
try:
    rows = list(cur.execute("select name,duration,grid_x,grid_y,grid_z,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size from kernels"))
except sqlite3.OperationalError:   # older rocpd views without grid_z
    rows = list(cur.execute("select name,duration,grid_x,grid_y,1,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size from kernels"))
agg = defaultdict(list)
for r in rows:
    agg[(r[0], r[2], r[3], r[4])].append(r)
print("\n# launch classes (kernel x grid), sorted by total time")
print("%-70s %8s %8s %6s %8s %10s %12s %6s %6s %8s" % ("kernel", "blocks_x", "grid_y", "grid_z", "calls", "avg_us", "total_us", "vgpr", "sgpr", "lds"))
for (name, gx, gy, gz), rs in sorted(agg.items(), key=lambda kv: -sum(r[1] for r in kv[1]))[:60]:
    tot = sum(r[1] for r in rs) / 1e3
    wg = max(rs[0][5], 1)
    print("%-70s %8d %8d %6d %8d %10.1f %12.1f %6d %6d %8d" % (name[:70], gx // wg, gy, gz, len(rs), tot / len(rs), tot, rs[0][6] + rs[0][7], rs[0][8], rs[0][9]))
