"""Fused Winograd kernel vs the three-launch Winograd path (GPU box): ms per layer for every 3x3 stride-1 layer class of
the 16 x 256^2 plan that is eligible for csrc/wino_fused.hip (variant 80 = fused, 81 = wino_input + component GEMMs +
wino_output, 0 = direct implicit GEMM).  usage: python tools/wino_fused_sweep.py [B] [filter] [deep] [pair]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
flt = sys.argv[2] if len(sys.argv) > 2 else ""
# (name, H, W, Cin, Cout, up, epi)
cases = [
    ("L0  64->64  film", 256, 256, 64, 64, 0, 1),
    ("L0  64->64  res", 256, 256, 64, 64, 0, 2),
    ("L0 128->64  film", 256, 256, 128, 64, 0, 1),
    ("L0 128->128 res", 256, 256, 128, 128, 0, 2),
    ("L0 192->128 film", 256, 256, 192, 128, 0, 1),
    ("L0 up 256->128", 128, 128, 256, 128, 1, 0),
    ("L1 128->128 film", 128, 128, 128, 128, 0, 1),
    ("L1 256->256 res", 128, 128, 256, 256, 0, 2),
    ("L1 384->256 film", 128, 128, 384, 256, 0, 1),
    ("L1 up 512->256", 64, 64, 512, 256, 1, 0),
    ("L2 256->256 film", 64, 64, 256, 256, 0, 1),
]
cases = [c for c in cases if flt in c[0]]
cases += [("L2 512->256 film", 64, 64, 512, 256, 0, 1), ("L2 768->512 film", 64, 64, 768, 512, 0, 1), ("L2 512->512 res", 64, 64, 512, 512, 0, 2),
          ("L3 512->512 res", 32, 32, 512, 512, 0, 2), ("L3 1024->1024", 32, 32, 1024, 1024, 0, 2)] if "deep" in sys.argv else []
# 80: r02 fused kernel (32 tiles x 32 couts); 400: r03 (16 tiles x 64 couts); 401 / 402: 400 without weight / patch traffic; 403: short U ring
print("B=%d  %-20s %9s %9s %9s %9s %9s %9s %9s   %s" % (B, "layer", "fused32", "fused64", "64 noW", "64 noPatch", "64 ring12", "3-launch", "direct",
                                                         "TF/s executed (of 157.3): fused32 / fused64"))
if "pair" in sys.argv:   # the fp16-pair twin of the 64-cout kernel (404; 405: 18 instead of 12 weight units in flight) next to the f32 kernel (400) and its no-traffic twins
    print("B=%d  %-20s %9s %9s %9s %9s %9s   %s" % (B, "layer", "fused64", "f64 pair", "pair ring18", "64 noW", "64 noPatch", "f32-equivalent TF/s: fused64 / pair"))
    for name, H, W, Cin, Cout, up, epi in cases:
        if Cin % 64 or Cout % 64:
            continue
        exec_flops = 36 * 2.0 * B * ((H << up) // 4) * ((W << up) // 4) * Cin * Cout
        res = []
        for v in (400, 404, 405, 401, 402):
            ms = ctypes.c_double()
            rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, 3, 1, up, epi, 10, ctypes.byref(ms))
            res.append(ms.value if rc == 0 else float("nan"))
        print("      %-20s %9.4f %9.4f %9.4f %9.4f %9.4f   %.1f / %.1f" % ((name,) + tuple(res) + (exec_flops / res[0] / 1e9, exec_flops / res[1] / 1e9)), flush=True)
    sys.exit(0)
for name, H, W, Cin, Cout, up, epi in cases:
    Ho, Wo = H << up, W << up
    exec_flops = 36 * 2.0 * B * (Ho // 4) * (Wo // 4) * Cin * Cout
    res = []
    for v in (80, 400, 401, 402, 403, 81, 0):
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, 3, 1, up, epi, 10, ctypes.byref(ms))
        res.append(ms.value if rc == 0 else float("nan"))
    print("      %-20s %9.4f %9.4f %9.4f %9.4f %9.4f %9.4f %9.4f   %.1f / %.1f" % ((name,) + tuple(res) + (exec_flops / res[0] / 1e9, exec_flops / res[1] / 1e9)), flush=True)
