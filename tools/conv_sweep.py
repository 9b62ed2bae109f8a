"""Conv-kernel tuning sweep on the GPU box: per-layer-class TFLOP/s for each tile variant."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.lib()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,3,5".split(","))]
# (name, B, H, W, Cin, Cout, K, stride, up, epi)
cases = [
    ("L0 3x3 64->64 film", 16, 256, 256, 64, 64, 3, 1, 0, 1),
    ("L0 3x3 64->64 res", 16, 256, 256, 64, 64, 3, 1, 0, 2),
    ("L0 1x1 64->384", 16, 256, 256, 64, 384, 1, 1, 0, 0),
    ("L0 3x3 192->128", 16, 256, 256, 192, 128, 3, 1, 0, 1),
    ("L0 3x3 128->128 res", 16, 256, 256, 128, 128, 3, 1, 0, 2),
    ("L1 3x3 128->128", 16, 128, 128, 128, 128, 3, 1, 0, 1),
    ("L1 3x3 384->256", 16, 128, 128, 384, 256, 3, 1, 0, 1),
    ("L2 3x3 768->512", 16, 64, 64, 768, 512, 3, 1, 0, 1),
    ("L2 3x3 256->256", 16, 64, 64, 256, 256, 3, 1, 0, 1),
    ("L3 3x3 512->512", 16, 32, 32, 512, 512, 3, 1, 0, 1),
    ("mid 3x3 1024->1024", 16, 32, 32, 1024, 1024, 3, 1, 0, 1),
    ("up 3x3 1536->1024", 16, 32, 32, 1536, 1024, 3, 1, 0, 1),
    ("upconv 1024->512 x2", 16, 32, 32, 1024, 512, 3, 1, 1, 0),
    ("down 4x4s2 256->512", 16, 64, 64, 256, 512, 4, 2, 0, 0),
    ("1x1 1536->1024", 16, 32, 32, 1536, 1024, 1, 1, 0, 0),
]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cases = [c for c in cases if flt in c[0]]
print("%-24s" % "case" + "".join("%14s" % ("v%d TF/s" % v) for v in variants))
for name, B, H, W, Cin, Cout, K, stride, up, epi in cases:
    Ho = ((H << up) + 2 * (1 if K == 4 else K // 2) - K) // stride + 1
    flops = 2.0 * B * Ho * Ho * Cout * K * K * Cin
    row = "%-24s" % name
    for v in variants:
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, K, stride, up, epi, 10, ctypes.byref(ms))
        row += "%14s" % ("%.1f" % (flops / ms.value / 1e9) if rc == 0 else "ERR")
    print(row, flush=True)
