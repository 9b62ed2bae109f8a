"""Per-launch-group timing of one network evaluation inside the sampler (GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import image_restoration_sde_amd as P
from image_restoration_sde_amd import _lib
from oracle import irsde_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
arch = sys.argv[4] if len(sys.argv) > 4 else "unet"
if arch == "nafnet":
    m = P.ConditionalNAFNet(3, width=64, enc_blk_nums=[1, 1, 1, 28], middle_blk_num=1, dec_blk_nums=[1, 1, 1, 1])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.naf_synth_params(seed=0, img_channel=3, width=64, middle_blk_num=1,
                       enc_blk_nums=(1, 1, 1, 28), dec_blk_nums=(1, 1, 1, 1)).items()})
else:
    m = P.ConditionalUNet(3, 3, 64, depth=4)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.synth_params(seed=0).items()})
m.engine_flags = flags
m = m.to("cuda:0").eval()
lq, xT = O.synth_inputs(1, B, S, S)
sde = P.IRSDE(10, 100, "cosine", 0.005, device="cuda:0")
sde.set_model(m); sde.set_mu(torch.from_numpy(lq).cuda()); sde.profile = True
sde.reverse_sde(torch.from_numpy(xT).cuda(), T=3)
sde.reverse_sde(torch.from_numpy(xT).cuda(), T=5)
torch.cuda.synchronize()
pr = sde.last_profile()
print({k: round(v, 3) if v < 1e6 else "%.4g" % v for k, v in pr.items()})
buf = ctypes.create_string_buffer(1 << 18)
_lib.check(_lib.lib().irsde_op_profile(m.engine().h, buf, len(buf)))
tot = 0.0
for line in buf.value.decode().splitlines():
    ms = float(line.split()[0]); tot += ms
    extra = ""
    if "flops=" in line:
        fl = float(line.split("flops=")[1].split()[0])
        ex = float(line.split("exec=")[1].split()[0]) if "exec=" in line else fl
        extra = "  -> %.1f TF/s executed, %.1f algorithmic" % (ex / ms / 1e9, fl / ms / 1e9)
    print(line + extra)
print("total %.3f ms per evaluation" % tot)
