"""wino4_fused64_kernel with non-temporal hints (irsde_bench_conv 406: residual loads + output stores = production since r03,
407: + patch loads) next to the instance without hints (408): ms per layer.  Under `rocprofv3 --pmc FETCH_SIZE` (tools/pmc_kernel.py "wino4_fused64") the same
command gives the fabric reads per instance.  usage: python tools/fused_nt_probe.py [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.probes_lib()   # measurement variants live in the PROBES build (make -C image_restoration_sde_amd/csrc PROBES=1)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cases = [("L0 128->128 res", 256, 256, 128, 128, 0, 2), ("L0 192->128 film", 256, 256, 192, 128, 0, 1), ("L0 up 256->128", 128, 128, 256, 128, 1, 0),
         ("L1 256->256 res", 128, 128, 256, 256, 0, 2), ("L1 up 512->256", 64, 64, 512, 256, 1, 0), ("L0 64->64 res", 256, 256, 64, 64, 0, 2)]
print("%-20s %9s %9s %9s" % ("layer (B=16)", "408 none", "406 nt-epi", "407 nt-all"))
for name, H, W, Cin, Cout, up, epi in cases:
    res = []
    for v in (408, 406, 407):
        ms = ctypes.c_double()
        rc = L.irsde_bench_conv(v, 16, H, W, Cin, Cout, 3, 1, up, epi, iters, ctypes.byref(ms))
        res.append(ms.value if rc == 0 else float("nan"))
    print("%-20s %9.4f %9.4f %9.4f" % ((name,) + tuple(res)), flush=True)
