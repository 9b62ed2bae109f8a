"""Correctness of experimental conv tile variants vs the float64 oracle (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
variants = [int(v) for v in sys.argv[1].split(",")]
for name, case in T.CONV_CASES.items():
    B, C0, C1, H, W, Cout, K, stride, pad, in_shift, has_bias, has_film, silu, has_res = case
    if Cout < 128: continue
    rs = np.random.RandomState(1)
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, K, K)) / np.sqrt((C0 + C1) * K * K)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32) if has_bias else None
    film = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if has_film else None
    Ho = ((H << in_shift) + 2 * pad - K) // stride + 1
    Wo = ((W << in_shift) + 2 * pad - K) // stride + 1
    res = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if has_res else None
    ref = T.oracle_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res)
    errs = ["v%d %.2e" % (v, T.relerr(T.run_conv(x0, x1, w, bias, stride, pad, in_shift, film, silu, res, naive=100 + v), ref)) for v in variants]
    print("%-24s %s" % (name, errs), flush=True)
