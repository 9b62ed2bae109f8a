"""Where does the fused Winograd kernel's time go?  irsde_bench_conv variant 82 + dflags, dflags bits: 1 every input-patch
load reads zeros (out-of-range buffer offset: issued, but no memory traffic), 2 the same for the weight fragments, 4 / 8
producer / MFMA waves at s_setprio 2, 16 no transform arithmetic, 32 no LDS writes of V, 128 no MFMAs.
usage: python tools/wino_fused_isolate.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
combos = [("fused", 0), ("no patch", 1), ("no weights", 2), ("neither", 3), ("prodprio", 4), ("mfmaprio", 8), ("noVALU", 16), ("noLDSw", 32),
          ("noMFMA", 128), ("noMFMA,VALU", 144), ("noMFMA,LDSw", 160), ("noMFMA,VALU,LDSw", 176), ("onlybarriers", 179)]
print("B=%d %-16s" % (B, "layer") + "".join("%17s" % c[0] for c in combos) + "   ideal")
for name, H, W, Cin, Cout, up, epi in [("L0  64->64", 256, 256, 64, 64, 0, 1), ("L0 192->128", 256, 256, 192, 128, 0, 1),
                                       ("L1 384->256", 128, 128, 384, 256, 0, 1)]:
    Ho, Wo = H << up, W << up
    ideal = 36 * 2.0 * B * (Ho // 4) * (Wo // 4) * Cin * Cout / 157.3e12 * 1e3
    row = []
    for _, f in combos:
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(80 if f == 0 else 82 + f, B, H, W, Cin, Cout, 3, 1, up, epi, 10, ctypes.byref(ms))
        row.append(ms.value if rc == 0 else float("nan"))
    print("     %-16s" % name + "".join("%17.4f" % v for v in row) + "   %.4f" % ideal, flush=True)
