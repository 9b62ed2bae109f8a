"""Engine build time, single-image latency (reference usage: batch 1), graph vs eager at small sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import image_restoration_sde_amd as P
from oracle import irsde_oracle as O
t0 = time.time()
m = P.ConditionalUNet(3, 3, 64, depth=4)
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.synth_params(seed=0).items()})
m = m.to("cuda:0").eval()
t1 = time.time()
m.engine()
torch.cuda.synchronize()
print("module build %.1f s, engine build (pack + Winograd weight transforms + upload) %.1f s" % (t1 - t0, time.time() - t1), flush=True)
for (B, S) in ((1, 128), (1, 256), (1, 512), (4, 256)):
    lq, xT = O.synth_inputs(1, B, S, S)
    x, c = torch.from_numpy(xT).cuda(), torch.from_numpy(lq).cuda()
    for graph in (True, False):
        sde = P.IRSDE(10, 100, "cosine", 0.005, device="cuda:0")
        sde.set_model(m); sde.set_mu(c); sde.use_graph = graph
        sde.reverse_posterior(x, T=5)
        torch.cuda.synchronize(); t0 = time.time()
        sde.reverse_posterior(x)
        torch.cuda.synchronize(); dt = time.time() - t0
        print("B=%d %dx%d T=100 posterior graph=%d: %.3f s per batch -> %.2f img/s (%.2f ms per step)" % (B, S, S, graph, dt, B / dt, dt * 10), flush=True)
