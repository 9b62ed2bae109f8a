"""r04 sweep of the fused Winograd F(4x4,3x3) kernels on the layer classes the B=16 256x256 plan runs them on (GPU box):
r03's one-block-per-tile-group kernel (irsde_bench_conv 410 / 409 pair) against the persistent kernel (430; 431 / 432: weight
fragments / patch loads read zeros; 433: no non-temporal hint; 434: fp16 pairs).  Prints ms and executed TFLOP/s (36 component
GEMMs = direct / 4) and the fraction of the 157.3 TFLOP/s f32 MFMA roof.
usage: python tools/wino_fused64p_sweep.py [variants, default 410,430,431,432,433,409,434] [name filter] [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "410,430,431,432,433,409,434").split(",")]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
# (name, H = W of the INPUT, Cin, Cout, up, epi, launches of this class per network evaluation)
cases = [
    ("L0  64->64  film", 256, 64, 64, 0, 1, 3),
    ("L0  64->64  res", 256, 64, 64, 0, 2, 3),
    ("L0 128->64  film", 256, 128, 64, 0, 1, 1),
    ("L0 128->128 res", 256, 128, 128, 0, 2, 2),
    ("L0 192->128 film", 256, 192, 128, 0, 1, 2),
    ("L0 up 256->128", 128, 256, 128, 1, 0, 1),
    ("L1 128->128 film", 128, 128, 128, 0, 1, 4),
    ("L1 256->256 res", 128, 256, 256, 0, 2, 2),
    ("L1 384->256 film", 128, 384, 256, 0, 1, 2),
    ("L1 up 512->256", 64, 512, 256, 1, 0, 1),
    ("L2 256->256 film", 64, 256, 256, 0, 1, 4),
    ("L2 512->512 res", 64, 512, 512, 0, 2, 2),
    ("L3 512->512 res", 32, 512, 512, 0, 2, 4),
]
cases = [c for c in cases if flt in c[0]]
print("B=%d  %-20s" % (B, "layer") + "".join("%10s" % ("v%d ms" % v) for v in variants) + "   TF/s executed / frac of 157.3: " + " | ".join("v%d" % v for v in variants[:2]))
tot = [0.0] * len(variants)
for name, H, Cin, Cout, up, epi, n in cases:
    Ho = H << up
    ex = 2.0 * B * Ho * Ho * Cout * 9 * Cin / 4.0
    row, tf = "      %-20s" % name, []
    for i, v in enumerate(variants):
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, B, H, H, Cin, Cout, 3, 1, up, epi, 10, ctypes.byref(ms))
        row += "%10s" % ("%.4f" % ms.value if rc == 0 else "ERR")
        tot[i] += n * ms.value if rc == 0 else float("nan")
        tf.append(ex / ms.value / 1e9 if rc == 0 and ms.value > 0 else 0.0)
    print(row + "   " + " | ".join("%.1f / %.2f" % (t, t / 157.3) for t in tf[:2]), flush=True)
print("      %-20s" % "sum x launches" + "".join("%10.3f" % t for t in tot))
