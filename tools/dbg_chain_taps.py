"""Debug (GPU box): per-tap error of the fp16-mode NAFNet (chain kernel on) against the oracle's fp16-operand restatement, and the chain
levels re-derived from the engine's own input taps.  usage: python tools/dbg_chain_taps.py [enc3] [lens]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O
enc3 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lens = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
def relerr(a, b): return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
rs = np.random.RandomState(5 + enc3)
cls = P.latent_bokeh.ConditionalNAFNet if lens else P.ConditionalNAFNet
encs = (1, 1, 1, enc3)
bp = O.naf_synth_params(seed=11, img_channel=4, width=64, middle_blk_num=1, enc_blk_nums=encs, dec_blk_nums=(1, 1, 1, 1), lens=lens)
for k in bp:
    if k.endswith(".beta") or k.endswith(".gamma"):
        bp[k] = (0.5 * rs.standard_normal(bp[k].shape)).astype(np.float32)
B = 2
xt = rs.standard_normal((B, 4, 64, 64)).astype(np.float32)
cond = rs.standard_normal((B, 4, 64, 64)).astype(np.float32)
tvec = np.array([5, 60])
li = [rs.uniform(0.1, 1.0, B).astype(np.float32) for _ in range(3)]
taps = {}
with O.f16_convs():
    ref = O.nafnet_forward(bp, xt, cond, tvec, encs, 1, (1, 1, 1, 1), dtype=np.float64, taps=taps, lens_info=li if lens else None)
for tag, flags in (("chain", _lib.FLAG_FP16), ("layers", _lib.FLAG_FP16 | _lib.FLAG_NO_NAF_CHAIN)):
    m = cls(img_channel=4, width=64, enc_blk_nums=list(encs), middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in bp.items()}, strict=True)
    m.engine_flags = flags | _lib.FLAG_KEEP_ACTIVATIONS
    m = m.to("cuda:0").eval()
    args = (torch.from_numpy(xt).cuda(), torch.from_numpy(cond).cuda(), torch.from_numpy(tvec))
    y = (m(*args, lens_info=[torch.from_numpy(v) for v in li]) if lens else m(*args)).cpu().numpy()
    print("== %s: output vs oracle(f16 operands) %.3g" % (tag, relerr(y, ref)))
    for name, want in taps.items():
        got = m.debug_tap(name).numpy()
        print("   %-12s %s  err %.3g   max|got| %.3g max|want| %.3g" % (name, got.shape, relerr(got, want), np.abs(got).max(), np.abs(want).max()))
