"""Accuracy and speed of the opt-in Winograd F(4x4,3x3) mode vs the all-direct path (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O
import test_gpu_parity as T
DEV = "cuda:0"
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
print("== conv level (vs float64 oracle): direct / F(2,3) / F(4,3)")
rs = np.random.RandomState(3)
for (B, C0, C1, H, W, Cout, up, film, silu, res) in [(2, 128, 0, 16, 24, 128, 0, 1, 1, 0), (1, 512, 256, 8, 8, 512, 0, 1, 1, 0),
                                                       (2, 256, 0, 12, 20, 256, 0, 0, 1, 1), (1, 256, 0, 6, 10, 128, 1, 0, 0, 0),
                                                       (1, 1024, 512, 4, 4, 1024, 0, 1, 1, 0)]:
    x0 = rs.standard_normal((B, C0, H, W)).astype(np.float32)
    x1 = rs.standard_normal((B, C1, H, W)).astype(np.float32) if C1 else None
    w = (rs.standard_normal((Cout, C0 + C1, 3, 3)) / np.sqrt((C0 + C1) * 9)).astype(np.float32)
    bias = rs.standard_normal(Cout).astype(np.float32)
    fl = (0.3 * rs.standard_normal((1, 2 * Cout))).astype(np.float32) if film else None
    Ho, Wo = H << up, W << up
    r = rs.standard_normal((B, Cout, Ho, Wo)).astype(np.float32) if res else None
    ref = T.oracle_conv(x0, x1, w, bias, 1, 1, up, fl, silu, r)
    e = [rel(T.run_conv(x0, x1, w, bias, 1, 1, up, fl, silu, r, naive=n), ref) for n in (0, 2, 3)]
    print("C=%d+%d->%d %dx%d up=%d: %.2e  %.2e  %.2e" % (C0, C1, Cout, Ho, Wo, up, *e))

def mk(flags):
    m = P.ConditionalUNet(3, 3, 64, depth=4)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.synth_params(seed=0).items()})
    m.engine_flags = flags
    return m.to(DEV).eval()
md, m2, m4 = mk(_lib.FLAG_NO_WINOGRAD), mk(_lib.FLAG_NO_WINOGRAD_F43), mk(0)
print("== network level, B=4 128x128, t=50 (difference to the all-direct path)")
lq, xT = O.synth_inputs(7, 4, 128, 128)
x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
yd = md(x, c, 50).cpu().numpy()
print("F(2,3) default: %.2e   F(4,3): %.2e" % (rel(m2(x, c, 50).cpu().numpy(), yd), rel(m4(x, c, 50).cpu().numpy(), yd)))
print("== sampler level, B=4 128x128, T=100, injected noise (difference to the all-direct path)")
z = torch.from_numpy(O.synth_noise(7, 100, (4, 3, 128, 128))).to(DEV)
outs = {}
for name, m in (("direct", md), ("f23", m2), ("f43", m4)):
    sde = P.IRSDE(10, 100, "cosine", 0.005, device=DEV)
    sde.set_model(m); sde.set_mu(c); sde.injected_noise = z
    for mode in ("sde", "posterior"):
        torch.cuda.synchronize(); t0 = time.time()
        outs[name, mode] = sde.reverse_sde(x).cpu().numpy() if mode == "sde" else sde.reverse_posterior(x).cpu().numpy()
        print(name, mode, "%.2f s" % (time.time() - t0), "max|x0| %.1f" % np.abs(outs[name, mode]).max())
for mode in ("sde", "posterior"):
    print(mode, "F(2,3) rel %.2e   F(4,3) rel %.2e   (abs %.2e)" % (rel(outs["f23", mode], outs["direct", mode]), rel(outs["f43", mode], outs["direct", mode]),
                                                                  np.abs(outs["f43", mode] - outs["direct", mode]).max()))
