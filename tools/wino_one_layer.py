"""Run a few launches of the fused Winograd kernels on two layer classes of the B=16 256x256 plan (for rocprofv3 counter passes, GPU box).
usage: python tools/wino_one_layer.py [variants, default 430,432,448] [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "430,432,448").split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for name, H, Cin, Cout, up, epi in (("L0 128->128 res", 256, 128, 128, 0, 2), ("L2 512->512 res", 64, 512, 512, 0, 2)):
    for v in variants:
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, 16, H, H, Cin, Cout, 3, 1, up, epi, iters, ctypes.byref(ms))
        print("%-18s variant %d: rc %d, %.4f ms" % (name, v, rc, ms.value), flush=True)
