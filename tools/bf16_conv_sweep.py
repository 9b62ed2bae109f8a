"""bf16-MFMA convolution kernels at the layer shapes of the 16 x 256^2 plan (GPU box): ms per layer and TFLOP/s for
variant 62 (bf16 operands, fp32 activation storage) and 63 (bf16 operands and bf16 activation storage).
usage: python tools/bf16_conv_sweep.py [B] [filter]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
flt = sys.argv[2] if len(sys.argv) > 2 else ""
# (name, H, W, Cin, Cout, K, stride, up, epi)
cases = [
    ("L0 3x3  64->64  film", 256, 256, 64, 64, 3, 1, 0, 1),
    ("L0 3x3 128->64  film", 256, 256, 128, 64, 3, 1, 0, 1),
    ("L0 3x3 128->128 res", 256, 256, 128, 128, 3, 1, 0, 2),
    ("L0 3x3 192->128 film", 256, 256, 192, 128, 3, 1, 0, 1),
    ("L0 up  256->128", 128, 128, 256, 128, 3, 1, 1, 0),
    ("L1 3x3 256->256 res", 128, 128, 256, 256, 3, 1, 0, 2),
    ("L1 3x3 384->256 film", 128, 128, 384, 256, 3, 1, 0, 1),
    ("L2 3x3 512->512 res", 64, 64, 512, 512, 3, 1, 0, 2),
    ("L2 3x3 768->512 film", 64, 64, 768, 512, 3, 1, 0, 1),
    ("L3 3x3 1024->1024 res", 32, 32, 1024, 1024, 3, 1, 0, 2),
    ("L3 3x3 1536->1024 film", 32, 32, 1536, 1024, 3, 1, 0, 1),
    ("L0 1x1 128->384", 256, 256, 128, 384, 1, 1, 0, 0),
    ("L0 1x1  64->384", 256, 256, 64, 384, 1, 1, 0, 0),
    ("L0 1x1 192->128", 256, 256, 192, 128, 1, 1, 0, 0),
    ("L1 1x1 384->256", 128, 128, 384, 256, 1, 1, 0, 0),
    ("L1 4x4s2 64->128", 256, 256, 64, 128, 4, 2, 0, 0),
]
cases = [c for c in cases if flt in c[0]]
VARS = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "62,63").split(",")]
print("B=%d  %-26s " % (B, "layer") + " ".join("%7d ms %7s" % (v, "TF/s") for v in VARS))
for name, H, W, Cin, Cout, K, st, up, epi in cases:
    Ho, Wo = (H << up) // st, (W << up) // st
    flops = 2.0 * B * Ho * Wo * Cin * Cout * K * K
    res = []
    for v in VARS:
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, K, st, up, epi, 10, ctypes.byref(ms))
        res.append(ms.value if rc == 0 else float("nan"))
    print("      %-26s " % name + " ".join("%10.4f %7.1f" % (r, flops / r / 1e9) for r in res), flush=True)
