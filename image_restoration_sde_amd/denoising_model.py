"""`DenoisingModel` — inference half of the reference's model wrapper
(/root/reference/codes/config/deraining/models/denoising_model.py + base_model.py), the drop-in boundary
`codes/config/*/test.py` drives:

    model = create_model(opt); sde.set_model(model.model)
    model.feed_data(noisy_state, LQ, GT); model.test(sde, mode=..., save_states=False)
    visuals = model.get_current_visuals()

Only the inference surface exists here (feed_data / test / get_current_visuals / load / load_network);
the optimiser, EMA, LR schedule and checkpoint *saving* belong to training, which is out of scope
(SURVEY.md §2).  `self.model` is this package's `ConditionalUNet`, whose arithmetic runs in libirsde_hip.so.
"""
from collections import OrderedDict

import torch

from . import unet as _unet
from . import nafnet as _nafnet


def define_G(opt):
    """networks.define_G (deraining/models/networks.py:10-15): class looked up by name."""
    opt_net = opt["network_G"]
    name = opt_net["which_model_G"]
    cls = getattr(_nafnet, name, None) or getattr(_unet, name)
    return cls(**opt_net["setting"])


def create_model(opt, task="deraining"):
    """models.create_model (deraining/models/__init__.py:6-15).  `task`: the reference keeps one copy of the wrapper per
    task directory; deblurring / deshadow / inpainting / sisr differ from deraining only in `test()` (hard-coded
    reverse_sde, deblurring/models/denoising_model.py:150-157) -> `ReverseSDEDenoisingModel`."""
    if opt["model"] != "denoising":
        raise NotImplementedError("Model [{:s}] not recognized.".format(opt["model"]))
    if task in ("deblurring", "deshadow", "inpainting", "sisr"):
        return ReverseSDEDenoisingModel(opt)
    return DenoisingModel(opt)


class DenoisingModel:
    def __init__(self, opt):
        self.opt = opt
        # base_model.py:12: "cuda" whenever gpu_ids is given; this implementation has no CPU path at all
        self.device = torch.device("cuda")
        self.is_train = bool(opt.get("is_train", False))
        if self.is_train:
            raise NotImplementedError("training is out of scope of the MI355X sampler (SURVEY.md §2)")
        self.model = define_G(opt).to(self.device)
        self.load()

    # ---- denoising_model.py:121-125 ----
    def feed_data(self, state, LQ, GT=None):
        self.state = state.to(self.device)      # noisy_state
        self.condition = LQ.to(self.device)     # LQ
        if GT is not None:
            self.state_0 = GT.to(self.device)   # GT

    # ---- denoising_model.py:150-160 (deraining: mode switch; deblurring & co. hard-code 'sde') ----
    def test(self, sde=None, mode="posterior", save_states=False):
        sde.set_mu(self.condition)
        self.model.eval()
        with torch.no_grad():
            if mode == "sde":
                self.output = sde.reverse_sde(self.state, save_states=save_states)
            elif mode == "posterior":
                self.output = sde.reverse_posterior(self.state, save_states=save_states)
            elif mode == "ode":  # latent/stereo variants' perform_ode (latent_denoising_model.py:177-191)
                self.output = sde.reverse_ode(self.state, save_states=save_states)
        self.model.train()

    # ---- denoising_model.py:165-171 ----
    def get_current_visuals(self, need_GT=True):
        out_dict = OrderedDict()
        out_dict["Input"] = self.condition.detach()[0].float().cpu()
        out_dict["Output"] = self.output.detach()[0].float().cpu()
        if need_GT and hasattr(self, "state_0"):
            out_dict["GT"] = self.state_0.detach()[0].float().cpu()
        return out_dict

    # ---- denoising_model.py:191-195, base_model.py:92-105 ----
    def load(self):
        path = (self.opt.get("path") or {}).get("pretrain_model_G")
        if path is not None:
            self.load_network(path, self.model, (self.opt.get("path") or {}).get("strict_load", True))

    def load_network(self, load_path, network, strict=True):
        if hasattr(network, "module"):
            network = network.module
        load_net = torch.load(load_path, map_location="cpu")
        clean = OrderedDict()
        for k, v in load_net.items():  # strip DataParallel/DDP "module." prefixes
            clean[k[7:] if k.startswith("module.") else k] = v
        network.load_state_dict(clean, strict=strict)


class ReverseSDEDenoisingModel(DenoisingModel):
    """The wrapper of the deblurring / deshadow / inpainting / sisr task directories: `test(sde, save_states)` always runs
    `sde.reverse_sde` (deblurring/models/denoising_model.py:150-157); everything else is the deraining wrapper."""

    def test(self, sde=None, save_states=False):
        return DenoisingModel.test(self, sde, mode="sde", save_states=save_states)
