"""configs[4] latent pipeline: is the T-step loop bound by the host (hipGraphLaunch / kernel launches) or by the GPU?
Per batch size: hipGraph replay vs eager launches, time until the call returns to the host vs time until the stream is idle.
usage: python tools/latent_host_bound.py [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import image_restoration_sde_amd as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
base = [w for w in bench.SECONDARY if w["model"] == "latent"][0]
for batch in (64, 16, 4):
    w = dict(base); w["batch"] = batch; w["T"] = T
    wl = bench.Workload(P, w, dev, 0, 1, "weak")
    for graph in (True, False):
        wl.sde.use_graph = graph
        wl.one_step(); torch.cuda.synchronize()
        th = tg = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            wl.one_step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            th += (t1 - t0) / 3; tg += (t2 - t0) / 3
        print("B = %2d  %-6s host returns after %.1f ms, stream idle after %.1f ms (%.2f ms per step) -> %.1f img/s"
              % (batch, "graph" if graph else "eager", th * 1e3, tg * 1e3, tg * 1e3 / T, batch / tg))
