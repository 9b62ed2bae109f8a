"""Phase timeline of the fused Winograd kernel (irsde_bench_conv variant 82: per-wave shader-clock stamps, averaged over all
blocks).  usage: python tools/wino_fused_timeline.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for H, W, Cin, Cout, up, epi in [(256, 256, 64, 64, 0, 1), (256, 256, 192, 128, 0, 1), (128, 128, 256, 128, 1, 0), (128, 128, 384, 256, 0, 1)]:
    ms = ctypes.c_double()
    rc = L.irsde_bench_conv(82, B, H, W, Cin, Cout, 3, 1, up, epi, 1, ctypes.byref(ms))
    if rc:
        print("ERR", L.irsde_last_error())
    rc = L.irsde_bench_conv(80, B, H, W, Cin, Cout, 3, 1, up, epi, 10, ctypes.byref(ms))
    print("  whole launch: %.4f ms" % ms.value, flush=True)
