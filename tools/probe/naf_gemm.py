import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from image_restoration_sde_amd import _lib
L = _lib.lib()
for (H, Cin, Cout) in [(64, 512, 1024), (64, 512, 512), (128, 256, 512), (128, 256, 256), (256, 128, 256), (256,128,128), (512, 64, 128), (512, 64, 64)]:
    out = []
    for v in (0, 3, 50):
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, 8, H, H, Cin, Cout, 1, 1, 0, 0, 20, ctypes.byref(ms))
        fl = 2.0 * 8 * H * H * Cin * Cout
        out.append("v%d %.4f ms %.1f TF/s" % (v, ms.value, fl / ms.value / 1e9) if rc == 0 else "v%d fail" % v)
    print(H, Cin, Cout, " | ".join(out), flush=True)
