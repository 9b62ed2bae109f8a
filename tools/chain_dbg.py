"""r06 debug: the split NAFBlock chain inside irsde_sample (graph / eager), one part; prints whether the next call reports a co-residency timeout."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
sys.path.insert(0, ROOT)
from bench import synth_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
L = _lib.lib()
L.irsde_debug_force_chain_groups(G)
for graph in (False, True):
    model = P.latent_bokeh.ConditionalNAFNet(img_channel=4, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    model.load_state_dict(synth_state_dict(model, 0))
    model = model.to(dev).eval()
    model.set_compute_dtype("fp16")
    sde = P.IRSDE(max_sigma=50, T=100, schedule="cosine", eps=0.005, device=dev)
    sde.set_model(model); sde.seed = 7; sde.use_graph = graph
    rs = np.random.RandomState(1)
    mu = torch.from_numpy(rs.rand(B, 4, 64, 64).astype(np.float32)).to(dev)
    li = [torch.from_numpy(rs.uniform(0.1, 1.0, B).astype(np.float32)) for _ in range(3)]
    sde.set_mu(mu)
    for T in (2, 2, 10, 100):
        t0 = time.time()
        try:
            out = sde.reverse_sde(sde.noise_state(mu), T=T, lens_info=li)
            torch.cuda.synchronize()
            print("graph=%s T=%d ok %.3f s finite=%s" % (graph, T, time.time() - t0, bool(torch.isfinite(out).all())), flush=True)
        except Exception as ex:
            print("graph=%s T=%d FAILED after %.3f s: %s" % (graph, T, time.time() - t0, str(ex)[-60:]), flush=True)
