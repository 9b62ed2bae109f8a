"""Split-operand GEMM prototype (csrc/gemm_split.hip) on the deep Winograd layer classes of the 16 x 256^2 plan (GPU box):
ms of the 36 component GEMMs alone (native f32 gemm_zloop / 3 bf16 planes / 2 planes) and of the whole three-launch layer.
TF/s columns are EXECUTED Winograd FLOPs (36 * 2 * T * Cin * Cout) per second: fp32-equivalent throughput.
usage: python tools/split_gemm_bench.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_restoration_sde_amd import _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cases = [  # name, H, W, Cin, Cout
    ("L3 512->512", 32, 32, 512, 512), ("L3 512->1024", 32, 32, 512, 1024), ("L3 1024->1024", 32, 32, 1024, 1024), ("L3 1536->1024", 32, 32, 1536, 1024),
    ("L2 1024->512", 64, 64, 1024, 512), ("L2 768->512", 64, 64, 768, 512), ("L2 512->512", 64, 64, 512, 512),
    ("L1 512->256", 128, 128, 512, 256), ("L1 384->256", 128, 128, 384, 256), ("L1 256->256", 128, 128, 256, 256),
    ("L0 128->128", 256, 256, 128, 128),
]
def run(v, H, W, Cin, Cout):
    ms = ctypes.c_double()
    rc = L.irsde_bench_conv(v, B, H, W, Cin, Cout, 3, 1, 0, 2, 10, ctypes.byref(ms))
    return ms.value if rc == 0 else float("nan")
# GEMM variants: 421 native f32 (gemm_zloop), 423 / 422 three / two planes on the 128 x 128 plane-major prototype, 472 the engine's
# pair-interleaved LDS-DMA kernel (473 / 475 / 476: without global loads / MFMAs / output stores)
print("B=%d %-16s | GEMMs alone (ms): %7s %7s %7s %7s | TF/s-equiv: %6s %6s %6s %6s | pairs kernel without: %7s %7s %7s | whole layer (ms): %7s %7s" % (
    B, "layer", "f32", "x3 128", "x2 128", "x2 pairs", "f32", "x3", "x2 128", "pairs", "loads", "MFMAs", "stores", "f32 3-l", "fused64"))
for name, H, W, Cin, Cout in cases:
    fl = 36 * 2.0 * B * (H // 4) * (W // 4) * Cin * Cout
    g = [run(v, H, W, Cin, Cout) for v in (421, 423, 422, 472)]
    ab = [run(v, H, W, Cin, Cout) for v in (473, 475, 476)]
    w = run(81, H, W, Cin, Cout)
    f64 = run(400, H, W, Cin, Cout) if Cin <= 1024 else float("nan")
    print("     %-16s |                   %7.4f %7.4f %7.4f %7.4f |             %6.1f %6.1f %6.1f %6.1f |                        %7.4f %7.4f %7.4f |                   %7.4f %7.4f" % (
        (name,) + tuple(g) + tuple(fl / x / 1e9 for x in g) + tuple(ab) + (w, f64)), flush=True)
