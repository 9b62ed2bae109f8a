"""How does the torch-CPU port scale with threads on this host? (bounded: small input, one forward each)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import irsde_oracle as O
from oracle import torch_cpu_port as TP
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as ex:
    print("no cgroup cpu.max", ex)
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ")
params = {k: torch.from_numpy(v) for k, v in O.synth_params(seed=0, nf=64, depth=4).items()}
for size in (128, 256):
    lq, xT = O.synth_inputs(1234, 1, size, size)
    x, mu = torch.from_numpy(xT), torch.from_numpy(lq)
    for th in (8, 16, 32, 64, 128):
        torch.set_num_threads(th)
        with torch.no_grad():
            TP.unet_forward(params, x, mu, 50)
            t0 = time.time(); TP.unet_forward(params, x, mu, 50); dt = time.time() - t0
        print("size %d threads %d: %.3f s/forward" % (size, th, dt), flush=True)
        if dt > 20: break
