"""Experiment: does splitting the batch over two engines / streams overlap the HBM-bound kernels (Winograd transforms,
LayerNorm, attention) of one half with the MFMA-bound GEMMs of the other?  16 images as 1 x 16 vs 2 x 8 concurrently."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import image_restoration_sde_amd as P
from oracle import irsde_oracle as O

dev = torch.device("cuda", 0)
params = O.synth_params(seed=0)
def mk():
    m = P.ConditionalUNet(3, 3, 64, depth=4)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return m.to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lq, xT = O.synth_inputs(1234, 16, 256, 256)
mu, x = torch.from_numpy(lq).to(dev), torch.from_numpy(xT).to(dev)
def sde_for(m, mu_):
    s = P.IRSDE(max_sigma=10, T=100, schedule="cosine", eps=0.005, device=dev)
    s.set_model(m); s.set_mu(mu_); s.seed = 7; s.use_graph = True
    return s
m0 = mk(); s0 = sde_for(m0, mu)
for _ in range(2): y = s0.reverse_sde(x, T=T)
torch.cuda.synchronize(); t = time.time(); y = s0.reverse_sde(x, T=T); torch.cuda.synchronize()
t1 = time.time() - t
print("1 x 16: %.1f ms per step" % (t1 * 1e3 / T))
ma, mb = mk(), mk()
sa, sb = sde_for(ma, mu[:8]), sde_for(mb, mu[8:])
sb.image_offset = 8
sta, stb = torch.cuda.Stream(), torch.cuda.Stream()
def run2():
    with torch.cuda.stream(sta): ya = sa.reverse_sde(x[:8], T=T)
    with torch.cuda.stream(stb): yb = sb.reverse_sde(x[8:], T=T)
    return ya, yb
for _ in range(2): run2()
torch.cuda.synchronize(); t = time.time(); ya, yb = run2(); torch.cuda.synchronize()
t2 = time.time() - t
print("2 x 8 concurrent: %.1f ms per step  (speedup %.3f)" % (t2 * 1e3 / T, t1 / t2))
print("max diff vs single:", float((torch.cat([ya, yb]) - y).abs().max()))
with torch.cuda.stream(sta): 
    torch.cuda.synchronize(); t = time.time(); ya = sa.reverse_sde(x[:8], T=T); torch.cuda.synchronize()
print("1 x 8 alone: %.1f ms per step" % ((time.time() - t) * 1e3 / T))
