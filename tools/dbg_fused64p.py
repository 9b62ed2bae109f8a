import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import run_conv, oracle_conv, relerr
B, C0, H, W, Cout = 1, 64, 16, 16, 64
rs = np.random.RandomState(1)
x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
w = (rs.standard_normal((Cout, C0, 3, 3)) / np.sqrt(C0 * 9)).astype(np.float32)
ref = oracle_conv(x0, None, w, None, 1, 1, 0, None, 0, None)
for rep in range(3):
    got = run_conv(x0, None, w, None, 1, 1, 0, None, 0, None, naive=34)
    bad = np.abs(got - ref) > 1e-3
    print("rep", rep, "bad", int(bad.sum()))
    b, c, y, x = np.nonzero(bad)
    print("  channels:", sorted(set(c.tolist())))
    for cc in sorted(set(c.tolist()))[:6]:
        m = bad[0, cc]
        print("  c=%d rows/cols:" % cc, [(int(yy), "".join("X" if m[yy, xx] else "." for xx in range(W))) for yy in range(H) if m[yy].any()])
    yy, xx, cc = y[0], x[0], c[0]
    print("  got", got[0, cc, yy, xx], "ref", ref[0, cc, yy, xx], " other got:", got[0, cc, yy - 12, xx], got[0, cc, yy - 8, xx], got[0, cc, yy-4, xx])
    # does the wrong value equal some other reference value?
    hits = np.argwhere(np.abs(ref - got[0, cc, yy, xx]) < 1e-5)
    print("  ref positions equal to the wrong value:", hits[:5].tolist())
