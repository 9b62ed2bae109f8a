"""Direct (implicit-GEMM) layers on the PAIR kernels of csrc/conv_igemm.hip vs the native f32 kernels (GPU box): ms and TF/s (FLOPs
counted once per f32 product).  usage: python tools/pair_conv_sweep.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.lib()
def run(v, B, H, W, Cin, Cout, K, stride, epi=0):
    ms = ctypes.c_double()
    rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, K, stride, 0, epi, 10, ctypes.byref(ms))
    return ms.value if rc == 0 else float("nan")
cases = [  # name, B, H, W, Cin, Cout, K, stride
    ("unet 1x1 1536->1024 @32", 16, 32, 32, 1536, 1024, 1, 1), ("unet 1x1 768->512 @64", 16, 64, 64, 768, 512, 1, 1),
    ("unet 1x1 384->256 @128", 16, 128, 128, 384, 256, 1, 1), ("unet 1x1 192->128 @256", 16, 256, 256, 192, 128, 1, 1),
    ("unet 4x4s2 64->128 @256", 16, 256, 256, 64, 128, 4, 2), ("unet 4x4s2 128->256 @128", 16, 128, 128, 128, 256, 4, 2),
    ("unet 4x4s2 256->512 @64", 16, 64, 64, 256, 512, 4, 2), ("unet qkv 1024->384 @32", 16, 32, 32, 1024, 384, 1, 1),
    ("naf 1x1 512->1024 @64 B8", 8, 64, 64, 512, 1024, 1, 1), ("naf 1x1 512->512 @64 B8", 8, 64, 64, 512, 512, 1, 1),
    ("naf 1x1 256->512 @128 B8", 8, 128, 128, 256, 512, 1, 1), ("naf 1x1 128->256 @256 B8", 8, 256, 256, 128, 256, 1, 1),
    ("naf 1x1 64->128 @512 B8", 8, 512, 512, 64, 128, 1, 1), ("naf 1x1 1024->2048 @32 B8", 8, 32, 32, 1024, 2048, 1, 1),
]
print("%-28s %9s %9s %9s %9s | TF/s: %7s %7s %7s" % ("layer", "f32", "f16 pairs", "bf16 pairs", "no 256 tile", "f32", "f16 p", "no256"))
for name, B, H, W, Cin, Cout, K, st in cases:
    Ho = (H + 2 * (1 if K == 4 else K // 2) - K) // st + 1
    fl = 2.0 * B * Ho * Ho * Cout * Cin * K * K
    r = [run(v, B, H, W, Cin, Cout, K, st) for v in (0, 480, 481, 482)]
    print("%-28s %9.4f %9.4f %9.4f %9.4f |       %7.1f %7.1f %7.1f" % (name, r[0], r[1], r[2], r[3], fl / r[0] / 1e9, fl / r[1] / 1e9, fl / r[3] / 1e9), flush=True)
