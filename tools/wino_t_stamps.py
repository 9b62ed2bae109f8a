"""r06: cycle stamps of wino4_fused64t_kernel (irsde_bench_conv 465, PROBES build) on layer classes of the B=16 256x256 plan (GPU box).
usage: python tools/wino_t_stamps.py [B] [variants, default 465; 4650 / 4651 / 4652: no patch traffic / stores dropped / residual loads dropped]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "465").split(",")]
for name, H, Cin, Cout, up, epi in (("L0 64->64 film", 256, 64, 64, 0, 1), ("L0 128->128 res", 256, 128, 128, 0, 2), ("L1 384->256 film", 128, 384, 256, 0, 1),
                                    ("L2 512->512 res", 64, 512, 512, 0, 2), ("L3 512->512 res", 32, 512, 512, 0, 2)):
    for v in variants:
        ms = ctypes.c_double()
        print(name, "variant", v, flush=True)
        rc = L.irsde_bench_conv(v, B, H, H, Cin, Cout, 3, 1, up, epi, 1, ctypes.byref(ms))
        if rc:
            print("  rc", rc)
