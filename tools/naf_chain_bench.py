"""naf_chain_kernel alone (irsde_bench_naf_chain): ms per launch and per NAFBlock for the three register / ring variants, by block count and batch (GPU box).
usage: python tools/naf_chain_bench.py [variants, default 1,2,3]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
NAMES = {1: "x in registers, ring 8", 2: "x in L2, ring 16", 3: "x in L2, ring 32", 22: "2 groups per image", 24: "4 groups per image"}
print("MFMA floor per block: 1600 MFMAs x 16 cycles x 2 waves per SIMD = 51.2k cycles = 21 us at 2.4 GHz; weights 3.5 MB per block and work-group")
for nb, B in ((28, 64), (28, 8), (1, 64), (4, 64), (28, 16), (28, 32), (28, 128), (28, 256)):
    row = "blocks=%2d B=%3d :" % (nb, B)
    for v in variants:
        ms = ctypes.c_double()
        rc = L.irsde_bench_naf_chain(v, nb, B, 5, ctypes.byref(ms))
        row += "  [%s] %s" % (NAMES.get(v, v), ("%.3f ms = %.1f us / block" % (ms.value, 1e3 * ms.value / nb)) if rc == 0 else "ERR")
    print(row, flush=True)
