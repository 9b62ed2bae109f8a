"""Full-size accuracy of the Winograd modes vs the all-direct path: B=8 256x256, network + T=100 sampler."""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O
DEV = "cuda:0"
rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
B, S, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 256, 100
def mk(flags):
    m = P.ConditionalUNet(3, 3, 64, depth=4)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.synth_params(seed=0).items()})
    m.engine_flags = flags
    return m.to(DEV).eval()
lq, xT = O.synth_inputs(7, B, S, S)
x, c = torch.from_numpy(xT).to(DEV), torch.from_numpy(lq).to(DEV)
z = torch.from_numpy(O.synth_noise(7, T, (B, 3, S, S))).to(DEV)
res = {}
for name, flags in (("direct", _lib.FLAG_NO_WINOGRAD), ("f23", _lib.FLAG_NO_WINOGRAD_F43), ("f43", 0)):
    m = mk(flags)
    res[name, "fwd"] = m(x, c, 50).cpu().numpy()
    buf = ctypes.create_string_buffer(1 << 18)
    _lib.check(_lib.lib().irsde_plan_describe(m.engine().h, B, S, S, buf, len(buf)))
    d = buf.value.decode()
    print(name, "winograd F2 layers:", d.count("winograd F2"), " F4 layers:", d.count("winograd F4"), " direct conv launches:", d.count("conv M="))
    sde = P.IRSDE(10, T, "cosine", 0.005, device=DEV)
    sde.set_model(m); sde.set_mu(c); sde.injected_noise = z
    for mode in ("sde", "posterior", "ode"):
        torch.cuda.synchronize(); t0 = time.time()
        fn = {"sde": sde.reverse_sde, "posterior": sde.reverse_posterior, "ode": sde.reverse_ode}[mode]
        res[name, mode] = fn(x).cpu().numpy()
        print(name, mode, "%.2f s -> %.3f img/s" % (time.time() - t0, B / (time.time() - t0)), "max|x0| %.1f" % np.abs(res[name, mode]).max(), flush=True)
    del m, sde
    torch.cuda.empty_cache()
for key in ("fwd", "sde", "posterior", "ode"):
    print("%-9s  F(2,3) vs direct: rel %.2e abs %.2e    F(4,3) vs direct: rel %.2e abs %.2e" % (
        key, rel(res["f23", key], res["direct", key]), np.abs(res["f23", key] - res["direct", key]).max(),
        rel(res["f43", key], res["direct", key]), np.abs(res["f43", key] - res["direct", key]).max()))
