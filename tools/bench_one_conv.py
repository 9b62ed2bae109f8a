"""One convolution layer through irsde_bench_conv (GPU box) — the command tools/pmc_kernel.sh profiles.
usage: python tools/bench_one_conv.py variant B H W Cin Cout K stride up epi [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_restoration_sde_amd import _lib
a = [int(x) for x in sys.argv[1:]]
iters = a[10] if len(a) > 10 else 5
ms = ctypes.c_double()
rc = _lib.probes_lib().irsde_bench_conv(*a[:10], iters, ctypes.byref(ms))
print("rc=%d  %.4f ms" % (rc, ms.value))
