"""Cycle budget of the persistent fused Winograd kernel (irsde_bench_conv 435, STAMP twin) on the plan's layer classes (GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
cases = [("L0  64->64  film", 256, 64, 64, 0, 1), ("L0  64->64  res", 256, 64, 64, 0, 2), ("L0 128->128 res", 256, 128, 128, 0, 2), ("L0 192->128 film", 256, 192, 128, 0, 1),
         ("L0 up 256->128", 128, 256, 128, 1, 0), ("L1 256->256 res", 128, 256, 256, 0, 2), ("L2 512->512 res", 64, 512, 512, 0, 2)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
# stamp twins (r04): 435 production; 2000 every OPT bit; 2001 / 2002 no weight / patch traffic; 2003 patches from an L2-resident window (epilogues 1 / 2 of this list only)
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "435").split(",")]
NAMES = {2010: "halo kernel", 2011: "halo kernel, no weight traffic", 2012: "halo kernel, no halo traffic", 435: "production", 2000: "OPT = 15", 2001: "no weight traffic", 2002: "no patch traffic", 2003: "hot patches", 2004: "production"}
for name, H, Cin, Cout, up, epi in cases:
    if flt not in name:
        continue
    for var in variants:
        if var != 435 and epi == 0:
            continue
        print(name, "|", NAMES.get(var, var), flush=True)
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(var, 16, H, H, Cin, Cout, 3, 1, up, epi, 1, ctypes.byref(ms))
        sys.stdout.flush()
        if rc:
            print("  rc=%d" % rc)
