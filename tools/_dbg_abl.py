import ctypes, os, sys
sys.path.insert(0, "/root/repo")
from image_restoration_sde_amd import _lib
L = _lib.lib()
def run(v, H, W, Cin, Cout, B=16):
    ms = ctypes.c_double()
    rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, 3, 1, 0, 2, 10, ctypes.byref(ms))
    return ms.value if rc == 0 else float("nan")
for name, H, W, Cin, Cout in [("L3 1024->1024", 32, 32, 1024, 1024), ("L2 1024->512", 64, 64, 1024, 512), ("L1 512->256", 128, 128, 512, 256)]:
    r = [run(v, H, W, Cin, Cout) for v in (442, 461, 462, 463, 464)]
    print("%-16s 256x256: full %.4f  no-loads %.4f  no-LDS-stores %.4f  no-MFMA %.4f  no-output-stores %.4f" % ((name,) + tuple(r)), flush=True)
