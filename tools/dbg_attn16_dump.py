"""Dump the input / output taps of the fused attention blocks (attention-sensitive weights, tests/test_gpu_parity.py::test_fused_attention_block_vs_oracle) for offline
analysis of the 16-bit modes' roundings against the oracle restatement.  usage (GPU box): python tools/dbg_attn16_dump.py <out.npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O
BLOCKS = {"downs.0.2.": 4096, "downs.1.2.": 1024, "downs.2.2.": 256, "ups.2.2.": 1024, "ups.3.2.": 4096}
params = dict(O.synth_params(seed=0, nf=64, depth=4))
for pref, n in BLOCKS.items():
    params[pref + "fn.fn.to_out.0.weight"] = params[pref + "fn.fn.to_out.0.weight"] * np.float32(n)
lq, xT = O.synth_inputs(1234, 1, 64, 64)
out = {}
for dtype in ("bf16", "fp16"):
    m = P.ConditionalUNet(3, 3, 64, depth=4)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}, strict=True)
    m.set_compute_dtype(dtype)
    m.engine_flags |= _lib.FLAG_KEEP_ACTIVATIONS
    m = m.to("cuda:0").eval()
    m(torch.from_numpy(xT).cuda(), torch.from_numpy(lq).cuda(), 50)
    for pref in ("downs.0.2.", "downs.2.2."):
        out[dtype + "/" + pref + "in"] = m.debug_tap(pref[:-3] + ".1").numpy()
        out[dtype + "/" + pref + "out"] = m.debug_tap(pref[:-1]).numpy()
np.savez_compressed(sys.argv[1], **out)
print("saved", sys.argv[1], {k: v.shape for k, v in out.items()})
